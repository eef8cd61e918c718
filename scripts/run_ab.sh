mkdir -p gpurun_out
out=gpurun_out/ab.txt; : > $out
run() { timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'iter %.3f'%d['ms_per_step'], 'upd %.4f'%d['roofline']['avg_launch_ms'], 'value %.4g'%d['value'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if v>0.004})" >> $out; }
for rep in 1 2; do
run "prio0"
CRL_STREAM2_PRIO=1 run "prio-high"
done
run "c3" "--workload c3 --steps 5 --warmup 2"
echo done
