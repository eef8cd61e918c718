# Final measurement of the round: default bench (exact blocked Fisher-Yates), the bijection variant, rocprofv3 kernel stats,
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes), RCCL path with a forced 1-rank communicator.
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 900 python $R/bench.py > $R/gpurun_out/final/bench_n1.json 2> $R/gpurun_out/final/bench_n1.err
timeout 600 python $R/bench.py --shuffle bijection --no-cpu-baseline > $R/gpurun_out/final/bench_n1_bijection.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/prof_bench.json 2> $R/gpurun_out/final/prof.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gae_kernel|update_x3" --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gae_kernel|update_x3" --output-format csv -d $R/gpurun_out/final/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/bench_n1_rccl_forced.json 2>/dev/null
cd $R/gpurun_out/final
for d in prof pmc_fetch pmc_write; do
  for f in $(find $d -name "*kernel_stats.csv" -o -name "*counter_collection.csv" 2>/dev/null); do cp $f ${d}_$(basename $f); done
  rm -rf $d
done
echo done
