# Measurement set of a round: default bench, rocprofv3 kernel stats, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# passes), SQ counters of the update kernel, RCCL path with a forced 1-rank communicator, per-shard sizes. Run on the GPU box:
#   bash scripts/final_measure.sh <tag>      (results under gpurun_out/<tag>/)
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python $R/bench.py --kernel-breakdown --no-cpu-baseline > $O/bench_n1_breakdown.json 2> /dev/null
CRL_GEMM=x3 timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_n1_x3.json 2> /dev/null
timeout 300 python $R/bench.py --shuffle bijection --no-cpu-baseline > $O/bench_n1_bijection.json 2> /dev/null
timeout 300 python $R/bench.py --minibatches 1 --no-cpu-baseline > $O/bench_n1_minibatches1.json 2> /dev/null
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_n1_rccl_forced.json 2>/dev/null
for nt in 8192 16384 32768; do
  timeout 300 python3 $R/bench.py --total-envs $nt --no-cpu-baseline > $O/bench_n1_envs$nt.json 2>/dev/null
  timeout 300 python3 $R/bench.py --total-envs $nt --no-cpu-baseline --kernel-breakdown > $O/bench_n1_envs${nt}_breakdown.json 2>/dev/null
done
timeout 600 python3 $R/bench.py --workload c3 --steps 5 --warmup 2 > $O/bench_c3_n1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 > /dev/null 2> $O/prof_c3.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.log
KR="update_x2_kernel|gae_kernel|adv_bucket_sums|pack_records|rollout_cartpole"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KR" --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KR" --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
# calibration of FETCH_SIZE on a gather of whole 64-byte records with a known byte count: the permute pass of CRL_GATHER=0
CRL_GATHER=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "permute_records|update_x2_kernel" --output-format csv -d $O/calib_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CRL_GATHER=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "permute_records|update_x2_kernel" --output-format csv -d $O/calib_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
# the standalone GAE kernel (the loop fuses the scan into the rollout): same counters on CRL_GAE_FUSE=0
CRL_GAE_FUSE=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gae_kernel" --output-format csv -d $O/pmc_fetch_gae -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CRL_GAE_FUSE=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gae_kernel" --output-format csv -d $O/pmc_write_gae -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "update_x2_kernel|rollout_cartpole" --output-format csv -d $O/pmc_sq_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $O
for d in prof prof_c3 pmc_fetch pmc_write pmc_fetch_gae pmc_write_gae calib_fetch calib_write pmc_sq_*; do
  for f in $(find $d -name "*kernel_stats.csv" -o -name "*counter_collection.csv" 2>/dev/null); do cp $f ${d}_$(basename $f); done
  rm -rf $d
done
# multi-rank functional runs on this one GPU (peer all-reduce) — labelled shared_gpu, not scaling points
for n in 2 4 8; do
  timeout 600 python $R/bench.py --gpus $n --comm peer --share-gpu --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_shared_gpu_n$n.json
done
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --comm peer > $O/bench_n1_peer_forced.json 2>/dev/null
timeout 900 python $R/scripts/parity_margins.py > $O/parity_margins.json 2>/dev/null
echo done
