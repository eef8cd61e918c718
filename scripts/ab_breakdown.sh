# same-box A/B of library builds on the headline workload with the per-kernel breakdown: bash scripts/ab_breakdown.sh <variant|default> …
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2 3; do for v in "$@"; do
  if [ $v = default ]; then unset CRL_LIB_PATH; else export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v %.4g env-steps/s  %.3f ms/iter  update launch %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --kernel-breakdown ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v    breakdown %.3f ms/iter:' % d['ms_per_step'], {k: round(x, 3) for k, x in d['kernel_ms_per_step'].items()})"
done; done
