mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/testq.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value %.4g ms %.3f'%(d['value'],d['ms_per_step']), 'upd ms %.4f'%d['roofline']['avg_launch_ms'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/q.txt
done
echo done
