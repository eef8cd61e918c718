"""Turns the rocprofv3 outputs of scripts/final_measure.sh into the two committed summaries:
  profiles/<tag>_pmc_summary.json       per kernel: mean per launch of every counter collected
  profiles/<tag>_pmc_hbm_traffic.json   per kernel: FETCH_SIZE / WRITE_SIZE (KB per launch) and the factor FETCH_SIZE has to be
                                        multiplied with before it is compared with a byte count
The factor follows MI355X_MICROARCH.md (HBM section): 2 for wide coalesced streaming reads (16 B per lane: FETCH_SIZE tallies the
128-B requests at 64 B) — pack_records_kernel, whose byte count is known (36 B read per sample), confirms it (raw / known ≈ 0.49);
1 for gathers of whole 64-byte records, calibrated on permute_records_kernel (CRL_GATHER=0 passes: 4 + 64 B read per sample, raw /
known ≈ 0.96). The update kernel of the default build fetches its records through the permutation (one random 64-byte record per
sample and role), i.e. the second pattern. gae_kernel reads 4 B / 1 B per lane: uncalibrated, raw figure kept (factor 1).
Usage: python scripts/summarize_pmc.py gpurun_out/<tag> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 65536 * 128


def short(name):
    return name.split("(")[0].strip()


def collect(pattern):
    acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values per dispatch
    for f in glob.glob(os.path.join(src, pattern)):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


summary = {"note": "rocprofv3 --pmc passes of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (scripts/final_measure.sh), means per "
                   "launch. FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them (separate passes); how they compare with byte counts: "
                   "scripts/summarize_pmc.py. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.", "kernels": {}}
for kern, ctrs in collect("pmc_*counter_collection.csv").items():
    if "gather0" in kern:
        continue
    summary["kernels"][kern] = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in ctrs.items()}
# the calibration passes (CRL_GATHER=0) live in their own files
calib = collect("calib_*counter_collection.csv")
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)

traffic = {}
main = collect("pmc_fetch*counter_collection.csv")
mainw = collect("pmc_write*counter_collection.csv")
for kern in sorted(set(main) | set(mainw)):
    f = main.get(kern, {}).get("FETCH_SIZE"); w = mainw.get(kern, {}).get("WRITE_SIZE")
    if not f or not w:
        continue
    factor = 2.0 if "pack_records" in kern else 1.0
    pattern = ("16-B-per-lane streaming reads" if factor == 2.0 else
               "64-byte records gathered through the permutation" if "update_" in kern else "uncalibrated width: raw figure")
    traffic[kern] = {"FETCH_SIZE_KB_per_launch_mean": sum(f) / len(f), "WRITE_SIZE_KB_per_launch_mean": sum(w) / len(w),
                     "fetch_factor": factor, "fetch_x2_corrected": factor == 2.0, "pattern": pattern}
cal = {}
for kern, ctrs in calib.items():
    if "FETCH_SIZE" in ctrs:
        cal.setdefault(kern, {})["FETCH_SIZE_KB_per_launch_mean"] = sum(ctrs["FETCH_SIZE"]) / len(ctrs["FETCH_SIZE"])
    if "WRITE_SIZE" in ctrs:
        cal.setdefault(kern, {})["WRITE_SIZE_KB_per_launch_mean"] = sum(ctrs["WRITE_SIZE"]) / len(ctrs["WRITE_SIZE"])
for kern, rec in cal.items():
    if "permute_records" in kern and "FETCH_SIZE_KB_per_launch_mean" in rec:
        rec["known_read_KB_per_launch"] = 4 * B * (64 + 4) / 1024      # all four epochs in one launch
        rec["raw_over_known"] = rec["FETCH_SIZE_KB_per_launch_mean"] / rec["known_read_KB_per_launch"]
for kern, rec in traffic.items():
    if "pack_records" in kern:
        rec["known_read_KB_per_launch"] = B * 36 / 1024
        rec["raw_over_known"] = rec["FETCH_SIZE_KB_per_launch_mean"] / rec["known_read_KB_per_launch"]
traffic["_calibration_CRL_GATHER=0"] = cal
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
