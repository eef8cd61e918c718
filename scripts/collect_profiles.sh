# Copies the judged summaries of gpurun_out/<tag>/ (scripts/final_measure.sh) into profiles/ under the round's names.
TAG=${1:-final}
S=gpurun_out/$TAG
cp $S/generated/*.json profiles/ 2>/dev/null
cp $S/bench_n1.json profiles/${TAG}_bench_n1.json
for f in suite breakdown x3 bijection minibatches1 rccl_forced peer_forced envs8192 envs16384 envs32768 envs8192_breakdown envs8192_peer_forced envs8192_peer_forced_three_launch; do
  [ -s $S/bench_n1_$f.json ] && cp $S/bench_n1_$f.json profiles/${TAG}_bench_n1_$f.json
done
[ -s $S/suite.json ] && cp $S/suite.json profiles/${TAG}_suite.json
cp $S/bench_c2_n1.json profiles/${TAG}_bench_c2_n1.json; cp $S/bench_c3_n1.json profiles/${TAG}_bench_c3_n1.json
for n in 2 4 8; do [ -s $S/bench_shared_gpu_n$n.json ] && cp $S/bench_shared_gpu_n$n.json profiles/${TAG}_bench_shared_gpu_n$n.json; done
cp $S/prof_*kernel_stats.csv profiles/${TAG}_rocprof_kernel_stats.csv 2>/dev/null
for f in $S/prof_[0-9]*kernel_stats.csv; do [ -s "$f" ] && cp $f profiles/${TAG}_rocprof_kernel_stats.csv; done
for f in $S/prof_c3_*kernel_stats.csv; do [ -s "$f" ] && cp $f profiles/${TAG}_c3_rocprof_kernel_stats.csv; done
for f in $S/prof_envs8192_*kernel_stats.csv; do [ -s "$f" ] && cp $f profiles/${TAG}_rocprof_kernel_stats_envs8192.csv; done
cp $S/parity_margins.json profiles/${TAG}_parity_margins.json 2>/dev/null
cp $S/gae_sizes.txt profiles/${TAG}_gae_sizes.txt 2>/dev/null
[ -s $S/train_curve.json ] && cp $S/train_curve.json profiles/${TAG}_train_curve.json
[ -s $S/bench_a2c.json ] && cp $S/bench_a2c.json profiles/${TAG}_bench_a2c.json
[ -s $S/bench_dqn.json ] && cp $S/bench_dqn.json profiles/${TAG}_bench_dqn.json
[ -s $S/gae_beyond_cache.txt ] && { echo "# scripts/bench_gae_big.py (crl_gae_bench), from scripts/final_measure.sh"; cat $S/gae_beyond_cache.txt; } > profiles/${TAG}_gae_beyond_cache_final.txt
ls -la profiles/${TAG}_*
[ -s $S/generated/${TAG}_c3_pmc_summary.json ] || cp gpurun_out/${TAG}_c3pmc/generated/${TAG}_c3_pmc_summary.json profiles/ 2>/dev/null
# profiles/ must reproduce the bench line: the summaries derived from the kernel-stats CSV are checked against the CSV that was copied beside them
python3 - $TAG <<'PY'
import csv, json, sys
tag = sys.argv[1]
j = json.load(open(f"profiles/{tag}_rocprof_update_avg.json"))
row = next(r for r in csv.DictReader(open(f"profiles/{tag}_rocprof_kernel_stats.csv")) if "update_x2_kernel" in r["Name"])
ok = abs(float(row["AverageNs"]) - j["avg_ns"]) < 0.5 and int(row["Calls"]) == j["calls"]
print("rocprof_update_avg.json vs kernel_stats.csv:", "consistent" if ok else f"MISMATCH: json {j['avg_ns']} ns x {j['calls']}, csv {row['AverageNs']} ns x {row['Calls']}")
line = json.load(open(f"profiles/{tag}_bench_n1.json"))
fr = line["roofline"].get("avg_launch_ms_rocprof")
if fr is not None and abs(fr * 1e6 - float(row["AverageNs"])) > 0.5:
    ok = False; print(f"MISMATCH: bench line quotes {fr * 1e6:.1f} ns, the committed CSV {row['AverageNs']} ns")
sys.exit(0 if ok else 3)
PY
