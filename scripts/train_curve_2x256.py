#!/usr/bin/env python3
"""Functional proof of the 2x256 kernels over hundreds of iterations: `ppo(config)` on the on-device CartPole with a 2x256 actor / critic (the layer-wise path: the
register-stationary forward / backward / rollout of csrc/wide_rs.hpp by default, round 5's kernels with option wide_rs = 0), 1024 envs x 128 steps x 200 updates =
26 M env steps, eight seeds per flavour. The parity tests compare one or two iterations with the oracle; a run that learns CartPole says the kernels, the LDS-resident
env state of the rollout and the batched critic pass hold up over 3,200 optimiser steps. Not a benchmark.   python scripts/train_curve_2x256.py > profiles/<tag>_train_curve_2x256.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ARGS = sys.argv[1:]
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("tc", os.path.join(ROOT, "scripts", "train_curve.py"))
src = open(os.path.join(ROOT, "scripts", "train_curve.py")).read()
ns = {"__name__": "tc", "__file__": os.path.join(ROOT, "scripts", "train_curve.py")}
exec(compile(src[:src.index("flavours = {")], "train_curve.py", "exec"), ns)     # the run() helper only, not the CartPole 2x64 sweep
run, crl = ns["run"], ns["crl"]
nt, k, iters = 1024, 128, 200
out = {"workload": f"ppo() on the on-device CartPole, 2x256 actor / critic, {nt} envs x {k} steps x {iters} updates, reference hyper-parameters otherwise", "flavours": {}}
FLAV = [(a, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(","))) if "=" in a else (f"wide_rs = {a}", {"wide_rs": int(a)}) for a in ARGS] or [("wide_rs = 27 (register-stationary forward, backward with dW3, rollout: default)", {"wide_rs": 27}), ("wide_rs = 0 (round 5's kernels)", {"wide_rs": 0})]
for name, opts in FLAV:
    runs = {str(sd): run(nt, k, iters, sd, hidden=256, gae_mode=crl._lib.GAE_COMPAT, stale_obs=1, options=opts) for sd in range(1, 9)}
    finals = sorted(v["final"] for v in runs.values())
    out["flavours"][name] = {"per_seed": runs, "median_final": finals[len(finals) // 2], "min_final": finals[0], "max_final": finals[-1],
                             "seeds_reaching_475": sum(1 for v in runs.values() if v["best"] >= 475.0)}
print(json.dumps(out, indent=1))
