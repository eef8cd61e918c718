mkdir -p gpurun_out
python -m pytest tests/test_gpu_wide.py tests/test_golden_widen.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/ab_tests.log
out=gpurun_out/ab.txt; : > $out
run() { timeout 300 python bench.py --no-cpu-baseline $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'upd %.4f'%d['roofline']['avg_launch_ms'], 'value %.4g'%d['value'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if v>0.004}, 'loss %.6g'%d['last_iteration']['loss'])" >> $out; }
run "c3 chunk2048" "--workload c3 --steps 5 --warmup 2"
CRL_WIDE_CHUNK2=4096 run "c3 chunk4096" "--workload c3 --steps 5 --warmup 2"
CRL_WIDE_CHUNK2=8192 run "c3 chunk8192" "--workload c3 --steps 5 --warmup 2"
run "headline" "--steps 10"
echo done
