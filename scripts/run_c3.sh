#!/bin/bash
# C3 (obs 8 / act 4 / 2x256, 16384 envs) side measurement + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
python bench.py --workload c3 --steps 5 --warmup 2 > gpurun_out/c3/bench_c3.json 2> gpurun_out/c3/bench_c3.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3/prof -o c3 -- python3 bench.py --workload c3 --steps 2 --warmup 1 > gpurun_out/c3/prof_bench.json 2> gpurun_out/c3/prof.err
find gpurun_out/c3/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/c3/c3_kernel_stats.csv \;
find gpurun_out/c3/prof -type f ! -name "*kernel_stats.csv" -delete
