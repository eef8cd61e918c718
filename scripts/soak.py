#!/usr/bin/env python3
"""Soak: long runs of every loop, checking for non-finite parameters / losses."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cleanrl_jl_amd as crl
L = crl._lib
out = {}
# PPO CartPole, 1500 iterations at 4096 envs (786 M env steps)
a = crl.Agent(crl.PPOConfig(num_envs=4096, num_steps=128, total_timesteps=4096 * 128 * 1500), seed=2)
h = a.handle; h.env_reset()
for i in range(15):
    st = h.iterate(100)
    es = h.episode_stats()
    assert np.isfinite(h.read(L.F_PARAMS)).all() and all(np.isfinite(s["loss"]) for s in st)
out["ppo_cartpole_mean_return_last"] = es["return_sum"] / max(es["episodes"], 1)
a.close()
# C3 shape on the synthetic env, 60 iterations
a = crl.Agent(crl.PPOConfig(num_envs=4096, num_steps=128, total_timesteps=4096 * 128 * 60), obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
h = a.handle; h.env_reset()
for i in range(6):
    st = h.iterate(10)
    assert np.isfinite(h.read(L.F_PARAMS)).all() and all(np.isfinite(s["loss"]) for s in st)
out["c3_last_loss"] = st[-1]["loss"]
a.close()
# A2C 300 K steps, DQN 100 K steps
g = crl.A2CAgent(crl.A2CConfig(total_timesteps=300_000)); eps = []
while g.handle.env()[1] < 300_000:
    t, ts, e = g.handle.run_until_update(); eps += e
    if t == 0: break
assert np.isfinite(g.handle.read_params()).all()
out["a2c_mean_return_last50"] = float(np.mean([r for r, _, _ in eps[-50:]])); g.close()
d = crl.DQNAgent(crl.DQNConfig(total_timesteps=100_000)); eps = []
while d.handle.status()["global_step"] < 100_000:
    t, e, _ = d.handle.run(20_000); eps += e
    if t == 0: break
assert np.isfinite(d.handle.read_params()[0]).all()
out["dqn_mean_return_last50"] = float(np.mean([r for r, _, _, _ in eps[-50:]])); d.close()
print(json.dumps(out))
