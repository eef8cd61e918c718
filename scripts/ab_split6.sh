# rollout_split = 1 (three waves per tile: rollout_split3_kernel) vs 3 (six waves: rollout_split6_kernel) at shard sizes on ONE box (verdict r5 item 4 ii):
#   bash scripts/ab_split6.sh [rounds] → per run: envs, flavour, env-steps/s, ms per iteration, rollout ms per iteration (events around the kernel: --kernel-breakdown)
n=${1:-3}
for nt in 4096 8192 16384; do for rep in $(seq 1 $n); do for v in 1 3; do
python bench.py --no-cpu-baseline --no-extras --total-envs $nt --steps 40 --warmup 5 --opt rollout_split=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nt $nt rollout_split=$v', '%.4g'%d['value'], 'ms/iter %.3f'%d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-extras --total-envs $nt --steps 20 --warmup 5 --kernel-breakdown --opt rollout_split=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('   breakdown: rollout %.3f ms/iter = %.2f us per step' % (d['kernel_ms_per_step']['rollout'], d['kernel_ms_per_step']['rollout']*1e3/128))"
done; done; done
