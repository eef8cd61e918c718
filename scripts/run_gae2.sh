mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "gae" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > gpurun_out/test12.log
for w in 0 1 2; do CRL_GAE_WIDE=$w timeout 120 python scripts/bench_gae.py 2>&1 | sed "s/^/wide=$w /" >> gpurun_out/gae_sweep2.txt; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gae -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
echo done
