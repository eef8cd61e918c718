# update_tile = 32 (update_x2_kernel) vs 16 (update_t16_kernel + the early-exit repair launch) at shard sizes on ONE box (verdict r5 item 4 i):
#   bash scripts/ab_tile16.sh [rounds] → one line per run: envs, tile, env-steps/s, ms per iteration, update µs per launch
n=${1:-3}
for nt in 4096 8192 16384 65536; do for rep in $(seq 1 $n); do for v in 32 16; do
python bench.py --no-cpu-baseline --no-extras --total-envs $nt --steps 40 --warmup 5 --opt update_tile=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nt $nt tile=$v', '%.4g'%d['value'], 'ms/iter %.3f'%d['ms_per_step'], 'update us/launch %.1f'%(1e3*(d['roofline']['avg_launch_ms'] or 0)))"
done; done; done
