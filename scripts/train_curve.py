#!/usr/bin/env python3
"""End-to-end sanity: PPO on the on-device CartPole with the reference's default hyper-parameters (only num_envs / num_steps
scaled up) — mean episode return per block of iterations. Not a benchmark; shows that the loop learns."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl

nt, k, iters = 1024, 128, 300
out = {}
for name, kw in (("compat (reference semantics: stale obs after reset, GAE slot k = 0)", dict(gae_mode=crl._lib.GAE_COMPAT, stale_obs=1)),
                 ("fixed (bootstrap GAE, fresh obs after reset)", dict(gae_mode=crl._lib.GAE_FIXED, stale_obs=0))):
    agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * iters), seed=1, **kw)
    h = agent.handle
    h.env_reset()
    curve = []
    acc_ret = acc_n = 0.0
    for it in range(iters):
        h.iterate(1, want_stats=False)
        es = h.episode_stats()
        acc_ret += es["return_sum"]; acc_n += es["episodes"]
        if (it + 1) % 30 == 0:
            curve.append(round(acc_ret / max(acc_n, 1.0), 1)); acc_ret = acc_n = 0.0
    out[name] = curve
    agent.close()
print(json.dumps({"config": f"num_envs={nt}, num_steps={k}, {iters} iterations ({nt*k*iters/1e6:.1f} M env steps), mean episode return per 30 iterations", "curves": out}, indent=1))
