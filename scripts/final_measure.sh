# Measurement set of a round, ONE script (run on the GPU box): bench lines, rocprofv3 kernel stats, PMC passes, parity margins.
#   bash scripts/final_measure.sh <tag>      → gpurun_out/<tag>/ ; then, back in the container:  bash scripts/collect_profiles.sh <tag>
# Order: profiler passes first, scripts/summarize_pmc.py on the box, then the bench lines (which read those summaries).
# Every PMC pass is its own rocprofv3 run with --pmc only (no trace domains), as the MI355X guide prescribes.
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
B="python3 $R/bench.py"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 > /dev/null 2> $O/prof_c3.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_bench.json 2> $O/prof.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_envs8192 -- python3 $R/bench.py --total-envs 8192 --no-cpu-baseline > /dev/null 2>&1
# --no-extras: the headline loop only — the default line's side records (configs.c2, configs.shard_8192, with_stats_readback) launch the same update
# kernel at other sizes, and a per-kernel average over all of them is no longer the headline's (first run of round 5: 45 M instead of 180 M instructions)
P="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
KR="update_x2_kernel|adv_bucket_sums|pack_records|rollout_cartpole"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d $O/pmc_fetch -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d $O/pmc_write -- $P > /dev/null 2>&1
# the standalone GAE kernel (the loop fuses the scan into the rollout): same counters over scripts/pmc_gae_write.py — a handle's crl_compute_gae launches with
# NOTHING on another stream (the TCC counters are device-wide: under `bench.py --opt gae_fuse=0` the scan was charged with the side stream's shuffle stores, round 4)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gae_kernel|gae_seg2_kernel" --output-format csv -d $O/pmc_fetch_gae -- python3 $R/scripts/pmc_gae_write.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gae_kernel|gae_seg2_kernel" --output-format csv -d $O/pmc_write_gae -- python3 $R/scripts/pmc_gae_write.py > /dev/null 2>&1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "update_x2_kernel|rollout_cartpole" --output-format csv -d $O/pmc_sq_$n -- $P > /dev/null 2>&1
done
cd $O
for d in prof prof_c3 prof_envs8192 pmc_fetch pmc_write pmc_fetch_gae pmc_write_gae pmc_sq_*; do
  for f in $(find $d -name "*kernel_stats.csv" -o -name "*counter_collection.csv" 2>/dev/null); do cp $f ${d}_$(basename $f); done
  rm -rf $d
done
# the summaries bench.py's roofline reads (stamped with the kernel-source hash) are made HERE, before the bench lines, so that the
# line of this very run carries them; a copy travels back in gpurun_out/<tag>/generated/ for scripts/collect_profiles.sh
cd $R && python3 scripts/summarize_pmc.py $O $TAG > $O/summarize_pmc.log 2>&1
# the fused 2x256 kernels' counters (incl. FETCH / WRITE per optimiser step) BEFORE the bench lines: configs.c3.roofline.traffic of the headline line reads them
cd $R && bash scripts/pmc_c3.sh $TAG > $O/pmc_c3.log 2>&1
mkdir -p $O/generated && cp profiles/${TAG}_c3_pmc_summary.json $O/generated/ 2>/dev/null
mkdir -p $O/generated && cp profiles/${TAG}_pmc_summary.json profiles/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_update_kernel_counts.json profiles/${TAG}_rocprof_update_avg.json $O/generated/
timeout 900 $B --strict-profiles > $O/bench_n1.json 2> $O/bench_n1.err 
timeout 900 $B --suite --no-cpu-baseline > $O/bench_n1_suite.json 2> $O/bench_n1_suite.err; cp $R/profiles/${TAG}_suite.json $O/suite.json 2>/dev/null
timeout 300 $B --kernel-breakdown --no-cpu-baseline > $O/bench_n1_breakdown.json 2> /dev/null
timeout 300 $B --opt gemm=1 --no-cpu-baseline > $O/bench_n1_x3.json 2> /dev/null
timeout 300 $B --shuffle bijection --no-cpu-baseline > $O/bench_n1_bijection.json 2> /dev/null
timeout 300 $B --minibatches 1 --no-cpu-baseline > $O/bench_n1_minibatches1.json 2> /dev/null
timeout 300 $B --opt comm_force=1 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n1_rccl_forced.json 2>/dev/null
timeout 300 $B --opt comm_force=1 --comm peer --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n1_peer_forced.json 2>/dev/null
# the 8-GPU job's per-rank shape with the exchange in the loop (1-rank mailbox: push to self): one-launch optimiser step, then the three-launch step
timeout 300 $B --opt comm_force=1 --comm peer --total-envs 8192 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_n1_envs8192_peer_forced.json 2>/dev/null
timeout 300 $B --opt comm_force=1 --opt fuse_optim=0 --comm peer --total-envs 8192 --steps 40 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_n1_envs8192_peer_forced_three_launch.json 2>/dev/null
for nt in 8192 16384 32768; do
  timeout 300 $B --total-envs $nt --no-cpu-baseline > $O/bench_n1_envs$nt.json 2>/dev/null
  timeout 300 $B --total-envs $nt --no-cpu-baseline --kernel-breakdown > $O/bench_n1_envs${nt}_breakdown.json 2>/dev/null
done
timeout 300 $B --workload c2 --steps 40 --no-cpu-baseline > $O/bench_c2_n1.json 2>/dev/null
timeout 600 $B --workload c3 --steps 5 --warmup 2 --regions 3 > $O/bench_c3_n1.json 2>/dev/null
timeout 900 python3 $R/scripts/parity_margins.py > $O/parity_margins.json 2>/dev/null
timeout 300 python3 $R/scripts/bench_gae_big.py > $O/gae_beyond_cache.txt 2>/dev/null
timeout 300 python3 $R/scripts/bench_gae.py 0 0 > $O/gae_sizes.txt 2>/dev/null
timeout 600 python3 $R/scripts/train_curve.py > $O/train_curve.json 2>/dev/null
timeout 300 python3 $R/scripts/product_error.py $TAG > $O/product_error.txt 2>&1; cp $R/profiles/${TAG}_product_error.json $O/generated/ 2>/dev/null
timeout 300 python3 $R/scripts/bench_a2c.py > $O/bench_a2c.json 2>/dev/null
timeout 300 python3 $R/scripts/bench_dqn.py > $O/bench_dqn.json 2>/dev/null
echo done
