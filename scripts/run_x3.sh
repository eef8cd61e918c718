mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/test7_x3.log
CRL_GEMM=f32 timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/test7_f32.log
for m in x3 f32; do
  CRL_GEMM=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m','value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd TF %.1f'%d['roofline']['achieved'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/x3_bench.txt
done
for st in 0 6; do for pct in 50 54; do
  CRL_X3_STAGGER=$st CRL_X3_ACTOR_PCT=$pct timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('x3 stagger $st pct $pct','value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd TF %.1f'%d['roofline']['achieved'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/x3_bench.txt
done; done
echo done
