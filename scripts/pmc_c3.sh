# PMC passes over the fused 2x256 kernels of BASELINE config 3 (bench.py --workload c3): forward, backward, weight gradient, rollout.
#   bash scripts/pmc_c3.sh <tag>  → gpurun_out/<tag>_c3pmc/*.csv and profiles/<tag>_c3_pmc_summary.json (stamped with the kernel-source hash)
# Each counter group is its own rocprofv3 run with --pmc only (no trace domains), as the MI355X guide prescribes. The SQ / TCC counters are
# device-wide, so the dW3 sweeps that normally co-run on the side stream are serialised for these passes (--opt shuffle_overlap=0): a kernel is
# charged with its own work only.
TAG=${1:-c3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${TAG}_c3pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --opt shuffle_overlap=0"
KR="wide_rs_fwd_kernel|wide_rs_bwd_kernel|wide_rs_rollout_kernel|wide_fused_fwd_pc_kernel|wide_fused_bwd_kernel|wide_wgrad_gen_kernel|wide_wgrad_split_kernel|wide_rollout_pc_kernel|wide_skinny_kernel"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$KR" --output-format csv -d $O/p_$n -- $P > /dev/null 2>&1
  for f in $(find $O/p_$n -name "*counter_collection.csv"); do cp $f $O/${n}.csv; done
  rm -rf $O/p_$n
done
cd $R && python3 - "$O" "$TAG" <<'PY'
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
O, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void crl::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
M, SIMDS = 16384 * 128 // 4, 1024
out = {"source_hash": bench.source_hash(),
       "command": "bench.py --workload c3 --steps 1 --warmup 1 --opt shuffle_overlap=0 under rocprofv3 --pmc (one counter group per pass; scripts/pmc_c3.sh)",
       "note": "means per launch; a launch of the three update-pass kernels covers BOTH networks at M = 524,288 samples. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count "
               "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (= 32 per v_mfma_f32_32x32x16_f16), GRBM_GUI_ACTIVE cycles summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KB "
               "(FETCH_SIZE tallies 128-B requests of 16-byte-per-lane streaming reads at 64 B: MI355X_MICROARCH.md)", "kernels": {}}
for k, c in sorted(acc.items()):
    rec = {n: {"mean": sum(v) / len(v), "launches": len(v)} for n, v in sorted(c.items())}
    d = {n: rec[n]["mean"] for n in rec}
    der = {}
    if "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8 * SIMDS          # SIMD-cycles of the launch
        if "SQ_ACTIVE_INST_VALU" in d: der["vector_pipe_busy"] = d["SQ_ACTIVE_INST_VALU"] * 4 / cyc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: der["matrix_pipe_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in d: der["both_busy"] = d["SQ_VALU_MFMA_COEXEC_CYCLES"] / cyc
        if "SQ_LDS_BANK_CONFLICT" in d: der["lds_bank_conflict_share_of_cu_cycles"] = d["SQ_LDS_BANK_CONFLICT"] / (d["GRBM_GUI_ACTIVE"] / 8 * 256)
    if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_INST_ANY" in d:
        der["waves_waiting_for_an_issue_slot"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in d: der["waves_parked_on_a_counter_or_barrier"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_ANY" in d: der["waves_issuing"] = d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_MFMA" in d and "SQ_INSTS_VALU" in d:
        der["valu_instructions_per_mfma"] = (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / max(d["SQ_INSTS_MFMA"], 1)
    rec["derived"] = der
    out["kernels"][k] = rec
# HBM traffic of ONE optimiser step (verdict r5 item 3: configs.c3.roofline.traffic): every update-pass kernel's FETCH_SIZE (x 2: 16-byte-per-lane streaming reads are
# tallied at half, MI355X_MICROARCH.md) + WRITE_SIZE, times its launches per step (launches ÷ the forward kernel's), against SURVEY §8(d)'s algorithmic 52 B per sample
# the update pass's forward: wide_rs_fwd_kernel<…, true> since round 6 (wide_fused_fwd_pc_kernel then only runs as the rollout's batched critic pass and is left out below)
fwd = next((k for k in out["kernels"] if "wide_rs_fwd_kernel" in k and "true" in k), None) or next((k for k in out["kernels"] if "wide_fused_fwd_pc_kernel" in k), None)
if fwd and "FETCH_SIZE" in out["kernels"][fwd]:
    steps = out["kernels"][fwd]["FETCH_SIZE"]["launches"]
    per = {}
    for k, rec in out["kernels"].items():
        if "rollout" in k or "FETCH_SIZE" not in rec or "WRITE_SIZE" not in rec or (k != fwd and "fwd" in k):
            continue
        lps = rec["FETCH_SIZE"]["launches"] / steps
        per[k] = {"launches_per_step": lps, "read_bytes": 2 * rec["FETCH_SIZE"]["mean"] * 1024 * lps, "written_bytes": rec["WRITE_SIZE"]["mean"] * 1024 * lps}
    out["per_step"] = {"kernels": per, "traffic_bytes": sum(v["read_bytes"] + v["written_bytes"] for v in per.values()),
                       "algorithmic_bytes": 52 * M, "note": "update-pass kernels of this PMC pass only (forward, backward, weight gradient, the two dW3 sweeps); the loss / reduce / optimiser "
                                                            "launches move < 1 % of it. algorithmic = 52 B per sample (obs 32 + action 4 + old logprob 4 + adv 4 + return 4 + old value 4) x M = 524,288"}
json.dump(out, open(os.path.join("profiles", f"{tag}_c3_pmc_summary.json"), "w"), indent=1)
for k, r in out["kernels"].items():
    print(k, json.dumps(r["derived"]))
PY
mkdir -p $O/generated && cp $R/profiles/${TAG}_c3_pmc_summary.json $O/generated/
