mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/test14.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ev.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ev -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_ev_prof.json 2>/dev/null
echo done
