R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -n 3
CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so timeout 200 python scripts/wstamps_pc.py 2>&1 | tail -n 3
for v in default pcprio0 default pcprio0; do
  if [ $v = default ]; then unset CRL_LIB_PATH; else export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
  timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v %.2f ms/iter update %.2f' % (d['ms_per_step'], d['kernel_ms_per_step']['update']))"
done
unset CRL_LIB_PATH
bash scripts/c3_kernels.sh default --no-extras 2>&1 | head -8
