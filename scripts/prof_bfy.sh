cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bfyprof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bfy_bench.json 2>/dev/null
cd $R/gpurun_out; cp $(find bfyprof -name "*kernel_stats.csv" | head -1) bfy_kernel_stats.csv; rm -rf bfyprof
