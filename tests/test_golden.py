"""Committed golden vectors (tests/golden/c1_fixture.npz, BASELINE config C1: num_envs=8, num_steps=128).
CPU: the oracle still reproduces them (freezes the restatement). GPU: the HIP path through the C ABI reproduces them
without running the oracle at all. Integer fields exact; floats |a-b| <= 1e-5*|b| + 1e-6."""
import os

import numpy as np
import pytest

import oraclelib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_fixture.npz"))
RTOL, ATOL = 1e-5, 1e-6


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= rtol * np.abs(b) + atol))


def test_oracle_reproduces_golden():
    cfg = O.make_config(num_envs=8, num_steps=128)
    params = G["params"].copy()
    a, lp, v, _ = O.get_action(cfg, params, G["act_obs"], G["act_u"])
    assert np.array_equal(a, G["act_action"]) and close(lp, G["act_logprob"]) and close(v, G["act_value"])
    for mode in (0, 1):
        adv, ret = O.gae_batch(G["gae_value"], G["gae_reward"], G["gae_terminal"], G["gae_next_value"], G["gae_next_done"],
                               0.99, 0.95, mode)
        assert np.array_equal(adv, G[f"gae_adv_mode{mode}"]) and np.array_equal(ret, G[f"gae_ret_mode{mode}"])
    st = O.State(cfg)
    st.params[:] = params
    st.env_init()
    stats = st.iterate(10, gen_perm=True)
    assert np.array_equal(st.action, G["it_action"]) and np.array_equal(st.perm, G["it_perm"])
    assert np.array_equal(st.obs, G["it_obs"]) and close(st.adv, G["it_adv"])
    got = np.array([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats])
    assert close(got, G["it_stats"], rtol=2e-5) and close(st.params, G["it_params_after"], rtol=1e-4)
    st.close()


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    return crl


@pytest.mark.gpu
def test_hip_get_action_and_gae_match_golden(crl):
    agent = crl.Agent(crl.PPOConfig(num_envs=8, num_steps=128), params=G["params"])
    a, lp, v = agent.handle.policy_act(G["act_obs"], G["act_u"])
    assert np.array_equal(a, G["act_action"]), "bit-exact action indices"
    assert close(lp, G["act_logprob"]) and close(v, G["act_value"])
    lp2, ent = crl.logprob_actions(G["act_obs"], agent.actor, (G["act_action"] + 1).astype(np.int32))
    assert close(lp2, G["act_logprob"]) and close(ent, G["act_entropy"])
    for mode in (0, 1):
        adv, ret = crl._lib.gae_host(G["gae_value"], G["gae_reward"], G["gae_terminal"], G["gae_next_value"],
                                     G["gae_next_done"], 0.99, 0.95, mode)
        assert np.array_equal(adv, G[f"gae_adv_mode{mode}"]) and np.array_equal(ret, G[f"gae_ret_mode{mode}"])
    agent.close()


@pytest.mark.gpu
def test_hip_full_iteration_matches_golden(crl):
    L = crl._lib
    agent = crl.Agent(crl.PPOConfig(num_envs=8, num_steps=128, total_timesteps=8 * 128 * 10), params=G["params"],
                      shuffle_mode=L.SHUFFLE_FISHER_YATES)
    h = agent.handle
    h.env_reset()
    stats = h.iterate(1)
    assert np.array_equal(h.read(L.F_ACTION), G["it_action"]) and np.array_equal(h.read(L.F_PERM), G["it_perm"])
    assert np.array_equal(h.read(L.F_OBS), G["it_obs"]) and np.array_equal(h.read(L.F_TERMINAL), G["it_terminal"])
    assert np.array_equal(h.read(L.F_REWARD), G["it_reward"])
    assert close(h.read(L.F_LOGPROB), G["it_logprob"]) and close(h.read(L.F_VALUE), G["it_value"])
    assert close(h.read(L.F_ADVANTAGE), G["it_adv"]) and close(h.read(L.F_RETURN), G["it_ret"])
    got = np.array([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats])
    assert close(got, G["it_stats"], rtol=2e-6)
    assert np.max(np.abs(h.read(L.F_PARAMS) - G["it_params_after"])) < 1e-6
    # gradient of the last minibatch at the final parameters (no optimiser step)
    h.adv_stats()
    st = h.update_minibatch(3, 0.0, apply_update=False)
    g = h.read(L.F_GRADS)
    off = O.param_offsets(O.make_config())
    for i in range(12):
        a, b = g[off[i]:off[i + 1]].astype(np.float64), G["grad_last_mb"][off[i]:off[i + 1]].astype(np.float64)
        assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-12), i
    assert abs(st["loss"] - G["grad_stats"][0]) <= 2e-6 * max(1.0, abs(G["grad_stats"][0]))
    agent.close()
