# same-box A/B of option values on the C3 workload:  bash scripts/ab_c3_opt.sh <option> <v1> <v2> …  (each measured twice, interleaved)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; o=$1; shift
for rep in 1 2; do for v in "$@"; do
  timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --opt $o=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$o=$v %.2f ms/iter update %.2f rollout %.2f' % (d['ms_per_step'], d['kernel_ms_per_step']['update'], d['kernel_ms_per_step']['rollout']))"
done; done
