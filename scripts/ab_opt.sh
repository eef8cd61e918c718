# A/B of ONE option on one box: bash scripts/ab_opt.sh <option> <rounds> [extra bench.py arguments]  → value / ms per iteration / update ms per launch
o=$1; n=$2; shift; shift
for r in $(seq 1 $n); do for v in 1 0; do
python bench.py --no-cpu-baseline --opt $o=$v "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$o=$v', '%.4g'%d['value'], '%.3f'%d['ms_per_step'], '%.4f'%(d['roofline']['avg_launch_ms'] or 0))"
done; done
