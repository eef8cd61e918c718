#!/usr/bin/env python3
"""The one-launch optimiser step counts arrivals on a 32-bit ticket that is never reset (144 per launch): it wraps after 29.8 M launches
(5 hours of training at 65536 envs). This runs a tiny configuration past the wrap and checks that training neither hangs nor diverges.
    python scripts/ticket_wrap_soak.py [iterations = 1_900_000]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cleanrl_jl_amd as crl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_900_000
cfg = crl.PPOConfig(num_envs=8, num_steps=8, total_timesteps=8 * 8 * n, anneal_lr=False, lr=1e-5)
a = crl.Agent(cfg, seed=3); h = a.handle
assert h.get_option("fuse_optim") == 1
h.env_reset()
t0 = time.time(); done = 0; chunk = 20000
launches_per_it = cfg.update_epochs * cfg.num_minibatches
while done < n:
    m = min(chunk, n - done)
    h.iterate(m, want_stats=False); h.sync(); done += m
    if done % 200000 == 0 or done == n:
        p = h.read(crl._lib.F_PARAMS)
        assert np.isfinite(p).all(), done
        print(json.dumps({"iterations": done, "fused_launches": done * launches_per_it, "ticket_arrivals_mod_2^32": (done * launches_per_it * 144) % 2**32,
                          "wrapped": done * launches_per_it * 144 >= 2**32, "seconds": round(time.time() - t0, 1)}), flush=True)
a.close()
