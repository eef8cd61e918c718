# first GPU pass of round 4: new tests, whole GPU suite, bench line, GAE beyond-cache sweep, shard-regime breakdowns
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -k "c4_shards" > $O/t_c4.log 2>&1; echo "c4 rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ppo_entry or libm or gae_matches" > $O/t_new.log 2>&1; echo "new rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
timeout 300 python scripts/bench_gae_big.py > $O/gae_big.txt 2>&1
timeout 200 python bench.py --total-envs 8192 --no-cpu-baseline --kernel-breakdown > $O/b8192_bd.json 2>/dev/null
timeout 200 python bench.py --total-envs 8192 --no-cpu-baseline > $O/b8192.json 2>/dev/null
timeout 200 python bench.py --workload c2 --steps 40 --no-cpu-baseline > $O/c2.json 2>/dev/null
timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/c3.json 2>/dev/null
tail -3 $O/t_c4.log $O/t_new.log $O/t_all.log; cat $O/summary.txt
