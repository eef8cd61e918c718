"""Parity of the on-device DQN loop (csrc/dqn.hip through the crl_dqn_* C ABI) against the CPU restatement of dqn.jl
(oracle/dqn_oracle.c) on the same seeds and initial weights. The path has no transcendental function and
both sides sum in the same order with contraction off, so the bar here is BIT-EXACT: trajectories, episode records, losses,
q-net and target-net parameters."""
import json
import os

import numpy as np
import pytest

import oraclelib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def crl():
    import cleanrl_jl_amd as crl
    assert crl.device_count() >= 1
    return crl


def test_q_values_match_oracle(crl):
    rng = np.random.default_rng(0)
    params = O.dqn_params(2) + (0.05 * rng.standard_normal(O.DQN_P)).astype(np.float32)
    agent = crl.DQNAgent(crl.DQNConfig(), params=params)
    obs = rng.standard_normal((4, 33))
    q = agent.handle.q_values(obs)
    ref = np.stack([O.dqn_forward(params, obs[:, i]) for i in range(33)], axis=1)
    assert np.array_equal(q, ref)
    agent.close()


@pytest.mark.parametrize("kw", [dict(), dict(batch_size=32, min_buff_size=40, train_freq=3, target_net_freq=9, buffer_size=150,
                                              epsilon_duration=300.0, log_frequencey=30)])
def test_run_matches_oracle_bit_for_bit(crl, kw):
    """Default DQNConfig, and a small ring (150 slots: wraps many times) with frequent updates / target copies."""
    T = 4000 if not kw else 1500
    params = O.dqn_params(1)
    cfg = crl.DQNConfig(total_timesteps=T, lr=1e-3, **kw)
    agent = crl.DQNAgent(cfg, params=params, seed=21)
    okw = {("log_frequency" if k == "log_frequencey" else k): v for k, v in kw.items()}
    st = O.DQNState(O.dqn_config(total_timesteps=T, lr=1e-3, seed=21, **okw), params)
    h = agent.handle
    total = 0
    for chunk in (1, 7, 200, 5, 1000, 100_000):
        tg, eps_g, ls_g = h.run(chunk)
        to, eps_o, ls_o = st.run(chunk)
        total += tg
        assert tg == to and eps_g == eps_o, "same trajectories, same episode records (return, length, step, ϵ)"
        assert ls_g == ls_o, "logged losses bit-exact"
        sg, so = h.status(), st.env()
        assert sg["global_step"] == so["global_step"] == total and sg["rb_size"] == so["rb_size"]
        assert sg["n_updates"] == so["n_updates"] and sg["last_loss"] == so["last_loss"]
        assert np.array_equal(sg["state"], so["state"])
        qg, tg_ = h.read_params(); qo, to_ = st.params()
        assert np.array_equal(qg, qo) and np.array_equal(tg_, to_)
    assert total == T and h.status()["n_updates"] > 100
    assert not np.array_equal(h.read_params()[0], params)
    agent.close(); st.close()


def test_dqn_entry_point_logs_reference_records(crl, tmp_path):
    cfg = crl.DQNConfig(run_name="t", total_timesteps=2500)
    agent = crl.dqn(cfg, seed=9, chunk=700, to_terminal=False, to_json=True, to_tensorboard=False, log_dir=str(tmp_path))
    assert agent.handle.status()["global_step"] == 2500
    recs = [json.loads(l) for l in open(os.path.join(tmp_path, "dqn|t.json"))]
    ep = [r for r in recs if r["msg"] == "Episode Statistics"]
    tr = [r for r in recs if r["msg"] == "Training Statistics"]
    assert ep and set(ep[0]) >= {"episode_return", "episode_length", "global_step", "ϵ", "steps_per_sec"}    # dqn.jl:88
    assert len(tr) == 2 and set(tr[0]) >= {"loss"}                                                           # steps 1000, 2000
    assert crl.linear_schedule(1.0, 0.05, 10_000, 5000) == pytest.approx(0.525)
    agent.close()


def test_dqn_errors(crl):
    with pytest.raises(crl.CrlError, match="min_buff_size"):
        crl.DQNAgent(crl.DQNConfig(min_buff_size=50))
    with pytest.raises(crl.CrlError, match="batch_size"):
        crl.DQNAgent(crl.DQNConfig(batch_size=5000))


def test_dqn_budget_and_total_timesteps_edges(crl):
    agent = crl.DQNAgent(crl.DQNConfig(total_timesteps=455), params=O.dqn_params(1), seed=4)
    h = agent.handle
    assert h.run(0)[0] == 0
    total = 0
    for chunk in (3, 207, 1000, 1000):
        total += h.run(chunk)[0]
    st = h.status()
    assert total == 455 and st["global_step"] == 455 and st["n_updates"] == len(range(210, 456, 10))
    ref = O.DQNState(O.dqn_config(total_timesteps=455, seed=4), O.dqn_params(1)); ref.run(10_000)
    assert np.array_equal(h.read_params()[0], ref.params()[0])
    agent.close(); ref.close()


def test_dqn_fresh_handle_refuses_to_run_and_library_init_is_reference_shaped(crl):
    """dqn.jl:39-40 must have happened: no run and no q-values on the zeros of a fresh handle; crl_dqn_init_params = make_nn's
    glorot-uniform layers in q_net AND target_net."""
    L = crl._lib
    agent = crl.DQNAgent(crl.DQNConfig(total_timesteps=500), params=L.dqn_make_nn_host(seed=2))
    cfg = agent.crl_cfg
    agent.close()
    h = L.DQNHandle(cfg, 0)
    with pytest.raises(crl.CrlError, match="parameters not set"):
        h.run(100)
    with pytest.raises(crl.CrlError, match="parameters not set"):
        h.q_values(np.zeros((4, 1)))
    h.init_params(2)
    q, t = h.read_params()
    assert np.array_equal(q, L.dqn_make_nn_host(seed=2)) and np.array_equal(q, t)
    h.run(100)
    h.close()
