mkdir -p gpurun_out
CRL_UPDATE=split python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gradient or full_iteration or many_minibatches or live_unclipped" 2>&1 | grep -E "passed|failed|error" > gpurun_out/pp_tests.log
out=gpurun_out/pp.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'upd_ms %.4f'%d['roofline']['avg_launch_ms'], 'iter %.3f'%d['ms_per_step'], 'value %.4g'%d['value'])" >> $out; }
export CRL_UPDATE=split; run "split"
CRL_DEBUG_ABLATE=1 run "split noprefetch"
CRL_DEBUG_ABLATE=0 run "split dbg0"
export CRL_UPDATE=pp; run "pp"
echo done
