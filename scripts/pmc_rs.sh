# PMC passes over the register-stationary 2x256 kernels (wide_rs.hpp): busy shares of the pipes and wave-cycle breakdown.
#   bash scripts/pmc_rs.sh "<bench.py options>" <kernel regex>     (each counter group its own rocprofv3 --pmc run, no trace domains)
R=${GRAFT_REPO_ROOT:-$PWD}
OPTS=$1; KR=$2
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --opt shuffle_overlap=0 $OPTS"
rm -rf /tmp/pmc_rs && mkdir -p /tmp/pmc_rs
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$KR" --output-format csv -d /tmp/pmc_rs/p_$n -- $P > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_rs/*/*/*counter_collection.csv") + glob.glob("/tmp/pmc_rs/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void crl::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    d = {n: sum(v) / len(v) for n, v in c.items()}
    print("==", k, "launches", len(next(iter(c.values()))))
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024
    if cyc:
        print("   launch %.1f us at 2.4 GHz-equivalent; vector pipe busy %.3f | matrix pipe busy %.3f (both at once %.3f) | LDS array busy %.3f (bank conflicts %.3f of it)" % (
            d["GRBM_GUI_ACTIVE"] / 8 / 2400.0, d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / cyc, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / cyc, d.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / cyc,
            d.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc / 4), d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   of wave cycles: waiting (any) %.3f | waiting on an instruction %.3f | issuing %.3f ; waiting on LDS %.3f" % (
            d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_WAIT_INST_LDS", 0) / wc))
    print("   instructions per launch: VALU %.3g MFMA %.3g LDS %.3g SALU %.3g VMEM rd %.3g wr %.3g" % tuple(d.get(n, 0) for n in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
PY
