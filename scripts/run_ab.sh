mkdir -p gpurun_out
out=gpurun_out/ab.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'upd %.4f'%d['roofline']['avg_launch_ms'], 'rollout %.3f'%d['kernel_ms_per_step']['rollout'])" >> $out; }
CRL_LIB_PATH=$PWD/cleanrl.jl_amd/variants/lib_head.so run "head"
for i in 5 6 7; do CRL_LIB_PATH=$PWD/cleanrl.jl_amd/variants/lib_b$i.so run "b$i"; done
CRL_LIB_PATH=$PWD/cleanrl.jl_amd/variants/lib_b6.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gradient or full_iteration or many_mini or live" 2>&1 | grep -E "passed|failed|rror" >> $out
echo done
