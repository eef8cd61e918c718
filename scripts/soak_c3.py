#!/usr/bin/env python3
"""Soak of the round-4 2x256 kernels (fused forward / backward / weight gradient, producer-consumer rollout): long runs at several
sizes — whole tiles only, fewer tiles than blocks, the C3 size — checking for non-finite parameters / losses and for the sticky
time-out words. python scripts/soak_c3.py > profiles/<tag>_soak_c3.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cleanrl_jl_amd as crl
L = crl._lib
out = {}
for nt, iters, shape in ((16384, 300, dict(obs_dim=8, n_act=4)), (4096, 300, dict(obs_dim=8, n_act=4)), (64, 300, dict(obs_dim=8, n_act=4)),
                         (1024, 200, dict(obs_dim=16, n_act=8)), (192, 200, dict(obs_dim=3, n_act=2)), (70, 100, dict(obs_dim=8, n_act=4))):
    a = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=128, total_timesteps=nt * 128 * iters), hidden=256, env_kind=L.ENV_SYNTHETIC, **shape)
    h = a.handle; h.env_reset()
    t0 = time.time(); losses = []
    for i in range(iters // 50):
        st = h.iterate(50)
        p = h.read(L.F_PARAMS)
        assert np.isfinite(p).all() and all(np.isfinite(s["loss"]) for s in st), (nt, i)
        losses.append(st[-1]["loss"])
    h.sync()
    out[f"nt{nt}_obs{shape['obs_dim']}_act{shape['n_act']}"] = {"iterations": iters, "env_steps": nt * 128 * iters, "seconds": round(time.time() - t0, 2),
                                                                 "loss_every_50": [round(x, 5) for x in losses], "param_abs_max": float(np.abs(p).max())}
    a.close()
print(json.dumps(out))
