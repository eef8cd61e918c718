"""Measured deviations of the HIP path from the CPU oracle for every quantity the parity tests bound (run on the GPU box):
the evidence behind the per-quantity bars in tests/ and DESIGN.md §1. Test infrastructure: uses the oracle as the checker."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cleanrl_jl_amd as crl  # noqa: E402
import oraclelib as O  # noqa: E402

F = crl._lib
out = {}


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))


# (a) forward: logprob / value on 4096 random observations
rng = np.random.default_rng(0)
cfg = O.make_config()
params = O.orthogonal_params(cfg, 3) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
agent = crl.Agent(crl.PPOConfig(num_envs=8, num_steps=128, total_timesteps=8 * 128 * 10), params=params)
obs = np.asfortranarray(rng.standard_normal((4, 4096)).astype(np.float32)); u = rng.random(4096)
a_o, lp_o, v_o, margin = O.get_action(cfg, params, obs, u)
a_g, lp_g, v_g = agent.handle.policy_act(obs, u)
same = a_g == a_o
out["policy_act"] = {"actions_differ": int((~same).sum()), "min_margin_of_differing": float(margin[~same].min()) if (~same).any() else None,
                     "logprob_max_rel": rel(lp_g[same], lp_o[same]), "value_max_abs": float(np.max(np.abs(v_g - v_o))),
                     "value_max_rel_where_|v|>0.01": rel(v_g[np.abs(v_o) > 0.01], v_o[np.abs(v_o) > 0.01])}
agent.close()

# (b) gradients: per-array relative L2 and loss scalars, three sizes
for nt, k in ((8, 128), (64, 128), (4096, 128)):
    cfgo = O.make_config(num_envs=nt, num_steps=k)
    params = O.orthogonal_params(cfgo, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfgo))).astype(np.float32)
    agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10), params=params)
    h = agent.handle
    st = O.State(cfgo); st.params[:] = params
    st.obs[:] = rng.standard_normal((4, nt, k)).astype(np.float32); st.action[:] = rng.integers(0, 2, (nt, k))
    st.logprob[:] = (np.log(0.5) + 0.3 * rng.standard_normal((nt, k))).astype(np.float32)
    st.value[:] = rng.standard_normal((nt, k)).astype(np.float32); st.adv[:] = (2 * rng.standard_normal((nt, k))).astype(np.float32)
    st.ret[:] = (10 * rng.standard_normal((nt, k))).astype(np.float32); st.perm[:] = rng.permutation(nt * k).astype(np.int32)
    for f, a in ((F.F_OBS, st.obs), (F.F_ACTION, st.action), (F.F_LOGPROB, st.logprob), (F.F_VALUE, st.value), (F.F_ADVANTAGE, st.adv),
                 (F.F_RETURN, st.ret), (F.F_PERM, st.perm)):
        h.write(f, a)
    h.adv_stats()
    M = nt * k // 4
    gs = h.update_minibatch(1, 0.0, apply_update=False)
    g = h.read(F.F_GRADS).astype(np.float64)
    go, so = O.loss_grad(cfgo, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[M:2 * M])
    off = O.param_offsets(cfgo)
    out[f"gradient_M{M}"] = {"per_array_rel_l2_max": max(float(np.linalg.norm(g[off[i]:off[i + 1]] - go[off[i]:off[i + 1]]) /
                                                               np.linalg.norm(go[off[i]:off[i + 1]])) for i in range(12)),
                             **{key + "_rel": abs(gs[key] - so[key]) / max(abs(so[key]), 1e-30) for key in ("loss", "pg_loss", "v_loss", "entropy_loss")}}
    agent.close(); st.close()

# (c) whole iterations: losses and parameters after 1..3 iterations (exact serial shuffle on both sides)
nt, k = 8, 128
agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10), shuffle_mode=F.SHUFFLE_FISHER_YATES)
params = agent.get_params()
cfgo = O.make_config(num_envs=nt, num_steps=k)
st = O.State(cfgo); st.params[:] = params; st.env_init()
agent.handle.env_reset()
it = []
for i in range(3):
    gs = agent.handle.iterate(1); os_ = st.iterate(10, gen_perm=True)
    it.append({"loss_max_rel": max(abs(a["loss"] - b["loss"]) / max(abs(b["loss"]), 1e-30) for a, b in zip(gs, os_)),
               "param_max_abs": float(np.max(np.abs(agent.handle.read(F.F_PARAMS) - st.params))),
               "param_rel_l2": float(np.linalg.norm(agent.handle.read(F.F_PARAMS).astype(np.float64) - st.params) / np.linalg.norm(st.params)),
               "actions_equal": bool(np.array_equal(agent.handle.read(F.F_ACTION), st.action))})
out["iterations"] = it
agent.close(); st.close()
print(json.dumps(out, indent=1))
