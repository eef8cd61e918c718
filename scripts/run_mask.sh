mkdir -p gpurun_out
for m in 2147483647 1023 65535; do
CRL_DEBUG_GATHER_MASK=$m timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('mask $m','value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd ms %.4f'%d['roofline']['avg_launch_ms'])" >> gpurun_out/mask.txt
done
echo done
