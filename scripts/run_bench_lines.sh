# The bench lines of scripts/final_measure.sh without the profiler passes. Usage: bash scripts/run_bench_lines.sh <tag>
TAG=${1:-lines}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 300 python $R/bench.py --kernel-breakdown --no-cpu-baseline > $O/bench_n1_breakdown.json 2> /dev/null
CRL_GEMM=x3 timeout 300 python $R/bench.py --no-cpu-baseline > $O/bench_n1_x3.json 2> /dev/null
timeout 300 python $R/bench.py --shuffle bijection --no-cpu-baseline > $O/bench_n1_bijection.json 2> /dev/null
timeout 300 python $R/bench.py --minibatches 1 --no-cpu-baseline > $O/bench_n1_minibatches1.json 2> /dev/null
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel-breakdown > $O/bench_n1_rccl_forced.json 2>/dev/null
CRL_COMM_FORCE=1 timeout 300 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel-breakdown --comm peer > $O/bench_n1_peer_forced.json 2>/dev/null
for nt in 8192 16384 32768; do
  timeout 300 python3 $R/bench.py --total-envs $nt --no-cpu-baseline > $O/bench_n1_envs$nt.json 2>/dev/null
  timeout 300 python3 $R/bench.py --total-envs $nt --no-cpu-baseline --kernel-breakdown > $O/bench_n1_envs${nt}_breakdown.json 2>/dev/null
done
timeout 600 python3 $R/bench.py --workload c3 --steps 5 --warmup 2 > $O/bench_c3_n1.json 2>/dev/null
for n in 2 4 8; do
  timeout 600 python $R/bench.py --gpus $n --comm peer --share-gpu --steps 10 --warmup 2 --no-cpu-baseline --kernel-breakdown 2>/dev/null | grep '^{' > $O/bench_shared_gpu_n$n.json
done
echo done
