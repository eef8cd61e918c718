"""Parity of the on-device A2C loop (csrc/a2c.hip through the crl_a2c_* C ABI) against the CPU restatement of a2c.jl
(oracle/a2c_oracle.c) on the same seeds and initial weights.
Bars: action indices, rewards, terminals, episode records bit-exact; the Float64 env state bit-exact (explicit polynomials
for sin/cos on both sides); losses within 1e-5 relative (observed ~1e-13: both sides are Float64, only exp/log differ)."""
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


def _params(seed=4):
    pc = O.make_config()
    p = O.orthogonal_params(pc, seed)
    off = O.param_offsets(pc)
    p[off[4]:off[5]] *= 20     # a policy that is not 50/50
    return p


def test_discounted_future_rewards_matches_oracle_and_reference_signature(crl):
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 513, 1012):
        r = (rng.random(n) > 0.1).astype(np.float64); t = (rng.random(n) < 0.05).astype(np.uint8)
        for last_terminal in (0, 1):
            t[-1] = last_terminal
            g = crl.discounted_future_rewards(r, t.astype(bool), 3.25, 0.99)
            assert np.array_equal(g, O.a2c_discounted_future_rewards(r, t, 3.25, 0.99))
    assert crl.discounted_future_rewards(np.array([1.0, 1.0, 0.0]), np.array([False, False, True]), 123.0, 0.5).tolist() == [1.5, 1.0, 0.0]
    assert crl.discounted_future_rewards(np.zeros(0), np.zeros(0, bool), 1.0, 0.9).shape == (0,)
    with pytest.raises(TypeError):
        crl.discounted_future_rewards(np.ones(3, np.float32), np.zeros(3, bool), 1.0, 0.9)   # a2c.jl:13: all one T
    with pytest.raises(ValueError):
        crl.discounted_future_rewards(np.ones(3), np.zeros(2, bool), 1.0, 0.9)


def test_collect_matches_oracle_step_for_step(crl):
    params = _params()
    cfg = crl.A2CConfig(total_timesteps=10_000)
    agent = crl.A2CAgent(cfg, params=params, seed=11)
    st = O.A2CState(O.a2c_config(total_timesteps=10_000, seed=11), params)
    h = agent.handle
    s_g, g_g, n_g = h.env(); s_o, g_o, n_o = st.env()
    assert np.array_equal(s_g, s_o) and g_g == g_o == 0 and n_g == n_o == 0
    total = 0
    for chunk in (1, 1, 3, 50, 200, 150):          # stops short of the first update (needs > 512 transitions)
        tg, ts_g, eps_g = h.run_until_update(max_env_steps=chunk)
        to, ts_o, eps_o = st.run_until_update(max_env_steps=chunk)
        total += chunk
        assert tg == to == chunk and not ts_g["trained"] and not ts_o["trained"]
        assert eps_g == eps_o
        s_g, g_g, n_g = h.env(); s_o, g_o, n_o = st.env()
        assert g_g == g_o == total and n_g == n_o
        assert np.array_equal(s_g, s_o), "Float64 CartPole is bit-exact by construction"
        bg, bo = h.buffer(), st.buffer()
        assert np.array_equal(bg[1], bo[1]), "action indices"
        assert np.array_equal(bg[0], bo[0]) and np.array_equal(bg[2], bo[2]) and np.array_equal(bg[3], bo[3])
    agent.close(); st.close()


def test_updates_match_oracle(crl):
    params = _params(5)
    cfg = crl.A2CConfig(total_timesteps=6000, lr=1e-3)
    agent = crl.A2CAgent(cfg, params=params, seed=3)
    st = O.A2CState(O.a2c_config(total_timesteps=6000, lr=1e-3, seed=3), params)
    h = agent.handle
    n_updates = 0
    while True:
        tg, ts_g, eps_g = h.run_until_update()
        to, ts_o, eps_o = st.run_until_update()
        assert tg == to and eps_g == eps_o, "same trajectories, same episode records"
        assert ts_g["trained"] == ts_o["trained"] and ts_g["n"] == ts_o["n"]
        if ts_g["trained"]:
            n_updates += 1
            for k in ("actor_loss", "critic_loss"):
                assert abs(ts_g[k] - ts_o[k]) <= 1e-5 * max(1.0, abs(ts_o[k])), (k, ts_g[k], ts_o[k])
                assert abs(ts_g[k] - ts_o[k]) <= 1e-9 * max(1.0, abs(ts_o[k])), "both sides are Float64"
            assert np.max(np.abs(h.read_params() - st.get_params())) < 1e-6
            assert h.env()[2] == 0, "Buffer.clear! after the update (a2c.jl:102)"
        if tg == 0 or h.env()[1] >= 6000:
            break
    assert n_updates >= 5 and h.env()[1] == st.env()[1] == 6000
    assert np.max(np.abs(h.read_params() - st.get_params())) < 1e-6
    assert not np.array_equal(h.read_params(), params)
    agent.close(); st.close()


def test_a2c_entry_point_logs_reference_records(crl, tmp_path):
    cfg = crl.A2CConfig(run_name="t", total_timesteps=3000)
    agent = crl.a2c(cfg, seed=9, to_terminal=False, to_json=True, to_tensorboard=False, log_dir=str(tmp_path))
    assert agent.handle.env()[1] == 3000
    recs = [json.loads(l) for l in open(os.path.join(tmp_path, "a2c|t.json"))]
    names = {r["msg"] for r in recs}
    assert names == {"Episode Statistics", "Training Statistics"}
    ep = next(r for r in recs if r["msg"] == "Episode Statistics")
    assert set(ep) >= {"episode_return", "episode_length", "global_step", "steps_per_sec"}     # a2c.jl:106
    tr = next(r for r in recs if r["msg"] == "Training Statistics")
    assert set(tr) >= {"actor_loss", "critic_loss"}                                             # a2c.jl:100
    # a2c.jl:100 comes before a2c.jl:106: an update's record precedes the record of the episode whose end triggered it, and that episode is
    # the first one whose buffer passed min_replay_size
    names_seq = [r["msg"] for r in recs]
    first = names_seq.index("Training Statistics")
    assert names_seq[first + 1] == "Episode Statistics"
    assert recs[first + 1]["global_step"] > cfg.min_replay_size >= (recs[first - 1]["global_step"] if first else 0)
    agent.close()


def test_a2c_fresh_handle_refuses_to_run_and_library_init_is_reference_shaped(crl):
    """a2c.jl:37 must have happened: a handle whose parameters were never set errors instead of training zeros; crl_a2c_init_params
    gives the orthogonal start of networks.jl:36-49."""
    L = crl._lib
    h = L.A2CHandle(L.CrlA2CConfig(1e-4, 2000, 512, 500, 0.99, 3), 0)
    with pytest.raises(crl.CrlError, match="parameters not set"):
        h.run_until_update()
    h.init_params(7)
    assert np.array_equal(h.read_params(), L.make_actor_critic_host(4, 2, 64, seed=7))
    taken, ts, _ = h.run_until_update()
    assert taken > 0
    h.close()


def test_a2c_errors(crl):
    with pytest.raises(crl.CrlError, match="min_replay_size"):
        crl.A2CAgent(crl.A2CConfig(min_replay_size=100))


def test_a2c_budget_and_total_timesteps_edges(crl):
    agent = crl.A2CAgent(crl.A2CConfig(total_timesteps=700), params=_params(), seed=2)
    h = agent.handle
    taken, ts, eps = h.run_until_update(max_env_steps=0)
    assert taken == 0 and not ts["trained"] and eps == []
    total = 0
    while True:
        taken, ts, eps = h.run_until_update(max_env_steps=300)
        total += taken
        if taken == 0:
            break
    assert total == 700 and h.env()[1] == 700          # the loop stops at total_timesteps (a2c.jl:53)
    assert h.run_until_update()[0] == 0
    agent.close()
