R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q > $O/t_wide.log 2>&1; echo "wide rc=$?" >> $O/summary.txt
for v in 2 1 0; do timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --opt wide_fuse=$v > $O/c3_fuse$v.json 2>$O/c3_fuse$v.err; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/prof_c3.log
cd $O; for f in $(find prof_c3 -name "*kernel_stats.csv"); do cp $f c3_kernel_stats.csv; done; rm -rf prof_c3
cat $O/summary.txt; tail -n 5 $O/t_wide.log
