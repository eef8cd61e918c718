mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/test8.log
for pct in 52 54 56 58; do
  CRL_X3_ACTOR_PCT=$pct timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('x3b pct $pct','value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd TF %.1f'%d['roofline']['achieved'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/x3b_bench.txt
done
echo done
