# same-box A/B of library builds on the C3 workload:  bash scripts/ab_c3.sh <variant|default> … (each measured three times, interleaved; STEPS=n overrides 8 iterations per run)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2 3; do for v in "$@"; do
  if [ $v = default ]; then unset CRL_LIB_PATH; else export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
  timeout 300 python bench.py --workload c3 --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$v %.2f ms/iter update %.2f rollout %.2f optim %.2f reduce %.2f' % (d['ms_per_step'], k['update'], k['rollout'], k['optim'], k['reduce']))"
done; done
