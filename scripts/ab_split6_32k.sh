for rep in 1 2; do
for o in "" "--opt rollout_split=3 --opt rollout_split_max_tiles=1024"; do
python bench.py --no-cpu-baseline --no-extras --total-envs 32768 --steps 30 --warmup 5 $o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nt 32768 [$o]', '%.4g'%d['value'], 'ms/iter %.3f'%d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-extras --total-envs 32768 --steps 20 --warmup 5 --kernel-breakdown $o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('   breakdown: rollout %.3f ms/iter = %.2f us per step' % (d['kernel_ms_per_step']['rollout'], d['kernel_ms_per_step']['rollout']*1e3/128))"
done; done
