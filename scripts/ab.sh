#!/bin/bash
# A/B of library builds on ONE box (boxes of the pool differ by up to 12 %): bash scripts/ab.sh <rounds> <variant> [<variant> …]
# "default" = the in-tree library, anything else = cleanrl.jl_amd/variants/<name>/libcleanrl_hip.so (scripts/build_variant.sh).
# Extra bench.py arguments through AB_ARGS. Prints value / ms per iteration / update-kernel ms per launch, one line per run.
R=${GRAFT_REPO_ROOT:-$PWD}
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then lib=$R/cleanrl.jl_amd/libcleanrl_hip.so; else lib=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
    CRL_LIB_PATH=$lib timeout 600 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d.get('kernel_ms_per_step',{})
print('$v', 'round $r', 'value %.4g' % d['value'], 'ms/iter %.3f' % d['ms_per_step'], 'update ms/launch %.4f' % (d['roofline']['avg_launch_ms'] or 0), {a: round(b,3) for a,b in k.items()})"
  done
done
