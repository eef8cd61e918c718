# per-kernel averages of the C3 update pass under rocprofv3 for one library build:  bash scripts/c3_kernels.sh <variant|default> [bench args]
R=${GRAFT_REPO_ROOT:-$PWD}; v=$1; shift
if [ "$v" = default ]; then lib=$R/cleanrl.jl_amd/libcleanrl_hip.so; else lib=$R/cleanrl.jl_amd/variants/$v/libcleanrl_hip.so; fi
export CRL_LIB_PATH=$lib
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$v && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/prof_$v.json 2>/dev/null
echo "== $v: $(python3 -c "import json;d=json.load(open('/tmp/prof_$v.json'));print('%.2f ms/iter, update %.2f' % (d['ms_per_step'], d['kernel_ms_per_step']['update']))")"
for f in $(find /tmp/prof_$v -name "*kernel_stats.csv"); do python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("   %-60s calls %4s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
done
