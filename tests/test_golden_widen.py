"""Committed golden vectors for the rows built after the PPO/CartPole path (tests/golden/widen_fixture.npz; generator:
tests/golden/make_golden_widen.py): BASELINE configs[2] (obs 8 / act 4 / 2x256), A2C, DQN.
CPU: the oracles still reproduce them. GPU: the HIP paths reproduce them through the C ABI without running an oracle."""
import os
import sys

import numpy as np
import pytest

import oraclelib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_widen import c3_inputs  # noqa: E402  (fixed-seed inputs only; no oracle call)

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "widen_fixture.npz"))


def probe_close(v, name, rtol, atol):
    v = np.asarray(v, np.float64)
    ok = np.all(np.abs(v[G[name + "_ix"]] - G[name + "_val"]) <= rtol * np.abs(G[name + "_val"]) + atol)
    return bool(ok) and abs(np.linalg.norm(v) - float(G[name + "_norm"])) <= 1e-5 * float(G[name + "_norm"])


def test_oracles_reproduce_golden():
    cfg, params = c3_inputs()
    st = O.State(cfg); st.params[:] = params; st.env_init()
    for it in range(2):
        stats = st.iterate(10, gen_perm=True)
    assert np.array_equal(st.action, G["c3_action"]) and np.array_equal(st.perm, G["c3_perm"])
    assert np.allclose([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats], G["c3_stats"], rtol=1e-6, atol=1e-9)
    assert probe_close(st.params, "c3_params", 1e-5, 1e-7)
    st.close()
    a = O.A2CState(O.a2c_config(total_timesteps=4000, lr=1e-3, seed=17), O.orthogonal_params(O.make_config(), 4))
    while True:
        taken, ts, e = a.run_until_update()
        if taken == 0 or a.env()[1] >= 4000:
            break
    assert np.array_equal(a.env()[0], G["a2c_env"]) and np.array_equal(a.get_params(), G["a2c_params"])
    a.close()
    d = O.DQNState(O.dqn_config(total_timesteps=1500, lr=1e-3, log_frequency=100, seed=23), O.dqn_params(1))
    d.run(10_000)
    assert np.array_equal(d.params()[0], G["dqn_q"]) and np.array_equal(d.params()[1], G["dqn_target"])
    d.close()


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    return crl


@pytest.mark.gpu
def test_hip_c3_iterations_match_golden(crl):
    cfg, params = c3_inputs()
    L = crl._lib
    agent = crl.Agent(crl.PPOConfig(num_envs=16, num_steps=32, total_timesteps=16 * 32 * 10), params=params, obs_dim=8, n_act=4,
                      hidden=256, env_kind=L.ENV_SYNTHETIC, shuffle_mode=L.SHUFFLE_FISHER_YATES)
    h = agent.handle
    h.env_reset()
    for it in range(2):
        stats = h.iterate(1)
    assert np.array_equal(h.read(L.F_ACTION), G["c3_action"]), "bit-exact action indices"
    assert np.array_equal(h.read(L.F_PERM), G["c3_perm"]) and np.array_equal(h.read(L.F_TERMINAL), G["c3_terminal"])
    assert np.array_equal(h.read(L.F_REWARD), G["c3_reward"])
    adv = h.read(L.F_ADVANTAGE)
    assert np.all(np.abs(adv - G["c3_adv"]) <= 1e-5 * np.abs(G["c3_adv"]) + 1e-6)
    got = np.array([[s["loss"], s["pg_loss"], s["v_loss"], s["entropy_loss"]] for s in stats])
    assert np.all(np.abs(got - G["c3_stats"]) <= 2e-5 * np.maximum(1.0, np.abs(G["c3_stats"])))
    assert probe_close(h.read(L.F_PARAMS), "c3_params", 1e-4, 2e-5)
    agent.close()


@pytest.mark.gpu
def test_hip_a2c_run_matches_golden(crl):
    agent = crl.A2CAgent(crl.A2CConfig(total_timesteps=4000, lr=1e-3), params=O.orthogonal_params(O.make_config(), 4), seed=17)
    eps, losses = [], []
    while True:
        taken, ts, e = agent.handle.run_until_update()
        eps += e
        if ts["trained"]:
            losses.append((ts["n"], ts["critic_loss"], ts["actor_loss"]))
        if taken == 0 or agent.handle.env()[1] >= 4000:
            break
    assert np.array_equal(np.array(eps, np.float64), G["a2c_episodes"]), "same trajectories: same episode records"
    assert np.allclose(np.array(losses), G["a2c_losses"], rtol=1e-9, atol=0)
    assert np.array_equal(agent.handle.env()[0], G["a2c_env"])
    assert np.max(np.abs(agent.handle.read_params() - G["a2c_params"])) < 1e-6
    agent.close()


@pytest.mark.gpu
def test_hip_dqn_run_matches_golden_bit_for_bit(crl):
    agent = crl.DQNAgent(crl.DQNConfig(total_timesteps=1500, lr=1e-3, log_frequencey=100), params=O.dqn_params(1), seed=23)
    taken, eps, losses = agent.handle.run(10_000)
    assert taken == 1500
    assert np.array_equal(np.array(eps, np.float64), G["dqn_episodes"]) and np.array_equal(np.array(losses, np.float64), G["dqn_losses"])
    q, t = agent.handle.read_params()
    assert np.array_equal(q, G["dqn_q"]) and np.array_equal(t, G["dqn_target"])
    assert np.array_equal(agent.handle.status()["state"], G["dqn_env"])
    agent.close()
