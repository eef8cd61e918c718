for r in 1 2; do
for o in 1 0; do
python bench.py --no-cpu-baseline --opt fuse_optim=$o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fuse_optim=$o', '%.4g'%d['value'], '%.3f'%d['ms_per_step'])"
python bench.py --no-cpu-baseline --workload c2 --steps 40 --opt fuse_optim=$o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('  c2 fuse_optim=$o', '%.4g'%d['value'], '%.3f'%d['ms_per_step'])"
done; done
