mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/ab_tests.log
out=gpurun_out/ab.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'upd %.4f'%d['roofline']['avg_launch_ms'], 'rollout %.3f'%d['kernel_ms_per_step']['rollout'], 'loss %.6g'%d['last_iteration']['loss'], 'ret %.4f'%d['last_iteration']['mean_episode_return'])" >> $out; }
for rep in 1 2; do
CRL_GEMM=x3 run "x3"
CRL_GEMM=x2 run "x2"
done
echo done
