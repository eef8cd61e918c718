mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/ab_tests.log
out=gpurun_out/ab.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'upd %.4f'%d['roofline']['avg_launch_ms'], 'value %.4g'%d['value'], 'loss %.6g'%d['last_iteration']['loss'])" >> $out; }
run "x2"; run "x2"
echo done
