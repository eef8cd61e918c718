"""Generates tests/golden/c1_fixture.npz with the CPU oracle (oracle/ppo_oracle.c).

The reference (sash-a/CleanRL.jl) is Julia with no tests, fixtures or runnable build here, so these vectors are produced by
the build's own restatement — they freeze the oracle (a regression in it is caught on CPU) and give the GPU box inputs and
expected outputs that do not depend on re-running the oracle. Shapes follow BASELINE config C1: num_envs=8, num_steps=128.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oraclelib as O  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    cfg = O.make_config(num_envs=8, num_steps=128)
    params = O.orthogonal_params(cfg, 7) + (0.03 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
    out = {"params": params}
    # get_action / critic on 96 observations (ppo.jl:21-32,128); draws with a margin to the CDF knots only
    obs = np.asfortranarray((rng.standard_normal((4, 96)) * np.array([[1.0], [1.0], [0.1], [1.0]])).astype(np.float32))
    u = rng.random(96)
    a, lp, v, margin = O.get_action(cfg, params, obs, u)
    assert (margin > 1e-4).all(), "regenerate: a draw sits on a CDF knot"
    lpa, ent = O.logprob_actions(cfg, params, obs, a)
    out.update(act_obs=obs, act_u=u, act_action=a, act_logprob=lp, act_value=v, act_entropy=ent)
    # gae rows (ppo.jl:48-73), both modes
    value = np.asfortranarray((rng.standard_normal((8, 128)) * 5).astype(np.float32))
    reward = np.asfortranarray((rng.random((8, 128)) > 0.03).astype(np.float32))
    term = np.asfortranarray((rng.random((8, 128)) < 0.03).astype(np.uint8))
    nv = rng.standard_normal(8).astype(np.float32); nd = (rng.random(8) < 0.3).astype(np.uint8)
    out.update(gae_value=value, gae_reward=reward, gae_terminal=term, gae_next_value=nv, gae_next_done=nd)
    for mode in (0, 1):
        adv, ret = O.gae_batch(value, reward, term, nv, nd, 0.99, 0.95, mode)
        out[f"gae_adv_mode{mode}"] = adv; out[f"gae_ret_mode{mode}"] = ret
    # one full iteration from a fixed state: rollout buffers, advantages, permutation, 16 loss records, parameters
    st = O.State(cfg)
    st.params[:] = params
    st.env_init()
    stats = st.iterate(10, gen_perm=True)
    out.update(it_action=st.action.copy(), it_obs=st.obs.copy(), it_reward=st.reward.copy(), it_terminal=st.terminal.copy(),
               it_logprob=st.logprob.copy(), it_value=st.value.copy(), it_adv=st.adv.copy(), it_ret=st.ret.copy(),
               it_perm=st.perm.copy(), it_params_after=st.params.copy(),
               it_stats=np.array([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats]))
    # the last minibatch's gradient at the FINAL parameters (loss closure + backward, ppo.jl:202-244)
    g, s = O.loss_grad(cfg, st.params.copy(), st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret,
                       st.perm[768:1024])
    out.update(grad_last_mb=g, grad_stats=np.array([s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]]))
    st.close()
    np.savez_compressed(os.path.join(HERE, "c1_fixture.npz"), **out)
    print("wrote", os.path.join(HERE, "c1_fixture.npz"), {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
