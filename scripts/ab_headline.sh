# same-box A/B of library builds on the headline workload (and the 8192-env shard): bash scripts/ab_headline.sh <variant|default> …
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2; do for v in "$@"; do
  if [ $v = default ]; then unset CRL_LIB_PATH; else export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v 65536: %.4g  %.3f ms/iter | readback loop %.3f ms' % (d['value'], d['ms_per_step'], d['with_stats_readback']['ms_per_step']))"
  timeout 300 python bench.py --no-cpu-baseline --no-extras --total-envs 8192 --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v  8192: %.4g  %.3f ms/iter' % (d['value'], d['ms_per_step']))"
done; done
