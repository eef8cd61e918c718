mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/test17.log
for sh in bijection blocked-fy; do
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --shuffle $sh 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$sh value %.4g ms %.3f'%(d['value'],d['ms_per_step']), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> gpurun_out/bfy.txt
done
echo done
