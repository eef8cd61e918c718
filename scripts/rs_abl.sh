#!/bin/bash
# Average durations of selected kernels of the C3 workload under rocprofv3, per library build (default or variants made by scripts/build_variant.sh): same-box A/Bs of wide_rs.hpp.
#   bash scripts/rs_abl.sh "<opt string>" <kernel regex> <variant> [<variant> …]   → one line per variant: average duration of the matching kernels
R=${GRAFT_REPO_ROOT:-$PWD}
OPTS=$1; KRE=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so
  [ "$v" = default ] && lib=$R/cleanrl.jl_amd/libcleanrl_hip.so
  rm -rf /tmp/rsabl_$v
  CRL_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rsabl_$v -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline $OPTS > /dev/null 2>&1
  f=$(find /tmp/rsabl_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" "$KRE" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]): print("   %-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
