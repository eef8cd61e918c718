mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/test4.log
for s in 0 3 6 9 12; do
  CRL_ROLLOUT_STAGGER=$s timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stagger',$s,'value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd TF %.1f'%d['roofline']['achieved'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/tune1.txt
done
echo done
