# update_prio_small sweep at shard sizes on one box
for nt in 8192 4096; do for rep in 1 2; do for v in 0 1 2 3; do
python bench.py --no-cpu-baseline --no-extras --total-envs $nt --steps 40 --warmup 5 --opt update_prio_small=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nt $nt prio=$v', '%.4g'%d['value'], 'ms/iter %.3f'%d['ms_per_step'], 'update us/launch %.1f'%(1e3*(d['roofline']['avg_launch_ms'] or 0)))"
done; done; done
