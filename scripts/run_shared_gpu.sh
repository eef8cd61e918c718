# Functional run of the multi-rank path on a 1-GPU box: all ranks time-share GPU 0 through the one-shot peer all-reduce
# (csrc/peer.hip). NOT a scaling measurement — the lines are labelled config.shared_gpu. Usage: bash scripts/run_shared_gpu.sh <tag>
tag=${1:-r02}
mkdir -p gpurun_out
for n in 2 4 8; do
  timeout 600 python bench.py --gpus $n --comm peer --share-gpu --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/shared_gpu_n$n.err \
    | grep '^{' > gpurun_out/${tag}_bench_shared_gpu_n$n.json
  echo "n=$n rc=$?"
done
timeout 1200 python -m pytest tests/test_gpu_dp2.py tests/test_gpu_dp.py -q -m gpu -rs 2>&1 | tail -8 > gpurun_out/${tag}_dp2_pytest.txt
cat gpurun_out/${tag}_dp2_pytest.txt
