mkdir -p gpurun_out/final
timeout 900 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final/prof_bench.json 2> $R/gpurun_out/final/prof.log
# HBM traffic of the GAE kernel: separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gae_kernel|update_x3" --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gae_kernel|update_x3" --output-format csv -d $R/gpurun_out/final/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/bench_n1_rccl_forced.json 2>/dev/null
echo done
