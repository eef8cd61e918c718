mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_dp.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" > gpurun_out/ab_tests.log
out=gpurun_out/ab.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'value %.4g'%d['value'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" >> $out; }
for rep in 1 2; do
CRL_SHUFFLE_OVERLAP=0 run "no-overlap"
CRL_SHUFFLE_OVERLAP=1 run "overlap"
done
CRL_SHUFFLE_OVERLAP=0 run "8192 no-overlap" "--total-envs 8192"
CRL_SHUFFLE_OVERLAP=1 run "8192 overlap" "--total-envs 8192"
CRL_COMM_FORCE=1 run "forced comm" 
echo done
