cd $GRAFT_REPO_ROOT
for pct in 50 51 52 53 54 55; do
  CRL_X3_ACTOR_PCT=$pct python bench.py --steps 10 --warmup 2 --no-cpu-baseline --shuffle bijection 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pct', $pct, round(d['ms_per_step'],3), round(d['kernel_ms_per_step']['update'],3))"
done
for st in 0 1 2 3 4 6; do
  CRL_X3_STAGGER=$st python bench.py --steps 10 --warmup 2 --no-cpu-baseline --shuffle bijection 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stagger', $st, round(d['ms_per_step'],3), round(d['kernel_ms_per_step']['update'],3))"
done
