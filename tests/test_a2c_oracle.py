"""Pins for the A2C oracle (oracle/a2c_oracle.c, restating a2c.jl): hand-derived known answers for
discounted_future_rewards (a2c.jl:13-24), an independent torch-float64 autograd restatement of the two loss closures
(a2c.jl:81-97), libm cross-checks of the Float64 env helpers, and loop-level invariants of a2c.jl:53-111."""
import math

import numpy as np
import pytest
import torch

import oraclelib as O


def test_discounted_future_rewards_known_answers():
    # a2c.jl:13-24 by hand: γ = 0.5, no terminal inside, last not terminal → bootstrap from final_value
    g = O.a2c_discounted_future_rewards([1, 1, 1], [0, 0, 0], 4.0, 0.5)
    assert g.tolist() == [1 + 0.5 * (1 + 0.5 * 3.0), 1 + 0.5 * 3.0, 1 + 0.5 * 4.0]
    # last transition terminal → its own reward is dropped too (future_rewards[1] = 0.0, a2c.jl:15)
    g = O.a2c_discounted_future_rewards([1, 1, 0], [0, 0, 1], 123.0, 0.5)
    assert g.tolist() == [1.5, 1.0, 0.0]
    # a terminal inside cuts the recursion and zeroes that slot (a2c.jl:20: t ? 0.0 : r + γ·next)
    g = O.a2c_discounted_future_rewards([1, 0, 1, 1], [0, 1, 0, 0], 2.0, 0.5)
    assert g.tolist() == [1.0, 0.0, 1 + 0.5 * 2.0, 1 + 0.5 * 2.0]
    assert O.a2c_discounted_future_rewards([], [], 1.0, 0.9).shape == (0,)
    assert O.a2c_discounted_future_rewards([2.0], [0], 3.0, 0.5).tolist() == [3.5]


def test_float64_helpers_against_libm():
    L = O.a2c_lib()
    for x in np.linspace(-0.45, 0.45, 181):
        assert abs(L.a2c_sin(x) - math.sin(x)) < 2e-16 and abs(L.a2c_cos(x) - math.cos(x)) < 2e-16
    for x in [-40.0, -5.0, -1.0, -0.2, -0.13, -1e-3, 0.0, 1e-9, 0.1, 0.1303, 0.1304, 0.5, 3.0, 29.9, 31.0, 400.0]:
        assert abs(L.a2c_tanh_fast(x) - math.tanh(x)) < 5e-12, x   # the polynomial branch is good to ~1e-13 for x² < 0.017
    # textbook CartPole dynamics in float64
    s = np.array([0.01, -0.02, 0.03, 0.04]); t = np.zeros(1, np.int32); done = np.zeros(1, np.int32)
    ref = s.copy()
    for i in range(20):
        a = (i // 3) % 2
        force = 10.0 if a == 1 else -10.0
        x, xd, th, thd = ref
        tmp = (force + 0.05 * thd * thd * math.sin(th)) / 1.1
        thacc = (9.8 * math.sin(th) - math.cos(th) * tmp) / (0.5 * (4.0 / 3.0 - 0.1 * math.cos(th) ** 2 / 1.1))
        xacc = tmp - 0.05 * thacc * math.cos(th) / 1.1
        ref = np.array([x + 0.02 * xd, xd + 0.02 * xacc, th + 0.02 * thd, thd + 0.02 * thacc])
        L.a2c_cartpole_step(s.ctypes.data_as(O.C.POINTER(O.C.c_double)), t.ctypes.data_as(O.C.POINTER(O.C.c_int32)), a, 500,
                            done.ctypes.data_as(O.C.POINTER(O.C.c_int32)))
        assert np.allclose(s, ref, rtol=0, atol=1e-14)
    assert t[0] == 20


def _torch_net(p, off, base, x, n_out, h, d):
    W1 = p[off[base]:off[base + 1]].reshape(d, h).T; b1 = p[off[base + 1]:off[base + 2]]
    W2 = p[off[base + 2]:off[base + 3]].reshape(h, h).T; b2 = p[off[base + 3]:off[base + 4]]
    W3 = p[off[base + 4]:off[base + 5]].reshape(h, n_out).T; b3 = p[off[base + 5]:off[base + 6]]
    h1 = torch.tanh(W1 @ x + b1[:, None]); h2 = torch.tanh(W2 @ h1 + b2[:, None])
    return W3 @ h2 + b3[:, None]


def test_loss_grads_match_torch_autograd():
    rng = np.random.default_rng(1)
    cfg = O.a2c_config()
    pc = O.make_config()
    off = O.param_offsets(pc)
    params = O.orthogonal_params(pc, 2) + (0.05 * rng.standard_normal(int(off[12]))).astype(np.float32)
    params[off[4]:off[5]] *= 20
    n = 77
    states = np.asfortranarray(rng.standard_normal((4, n)) * np.array([[1.0], [1.0], [0.1], [1.0]]))
    actions = rng.integers(0, 2, n).astype(np.int32)
    returns = rng.random(n) * 50
    g, cl, al, adv = O.a2c_loss_grads(cfg, params, states, actions, returns)
    p64 = torch.tensor(params.astype(np.float64), requires_grad=True)
    x = torch.tensor(states)
    v = _torch_net(p64, off, 6, x, 1, 64, 4)[0]
    advantage = torch.tensor(returns) - v
    closs = (advantage ** 2).mean()                                        # a2c.jl:84-86
    z = _torch_net(p64, off, 0, x, 2, 64, 4)
    logp = torch.log(torch.softmax(z, 0))[torch.tensor(actions.astype(np.int64)), torch.arange(n)]
    aloss = -(logp * advantage.detach()).mean()                            # a2c.jl:92-96 (advantage is a captured constant)
    (closs + aloss).backward()
    gt = p64.grad.numpy()
    assert abs(cl - closs.item()) < 1e-9 * abs(cl) and abs(al - aloss.item()) < 1e-9 * max(1, abs(al))
    assert np.allclose(adv, advantage.detach().numpy(), rtol=1e-9, atol=1e-9)   # exp-based tanh_fast vs torch.tanh: ~1e-12
    for i in range(12):
        a, b = g[off[i]:off[i + 1]].astype(np.float64), gt[off[i]:off[i + 1]]
        assert np.linalg.norm(a - b) <= 1e-6 * max(np.linalg.norm(b), 1e-9), i   # float32 projection of the gradient


def test_loop_invariants():
    """a2c.jl:53-111: training happens only at an episode end with size > min_replay_size; the buffer is cleared after;
    episode records carry the global step; the first update's losses equal loss_grads on the buffer seen before it."""
    cfg = O.a2c_config(min_replay_size=512, total_timesteps=5000, seed=3)
    pc = O.make_config()
    params = O.orthogonal_params(pc, 4)
    st = O.A2CState(cfg, params)
    # stop just before the update to look at the buffer: run step by step
    seen = 0
    while True:
        taken, ts, eps = st.run_until_update(max_env_steps=1)
        seen += taken
        _, gstep, size = st.env()
        assert gstep == seen
        if ts["trained"]:
            assert size == 0 and ts["n"] > 512 and eps and eps[-1][2] == gstep
            break
        assert size <= 512 + 501
    st.close()
    # replay: a fresh state reaches the same update; rebuild its inputs from the buffer one step earlier
    st = O.A2CState(cfg, params)
    st.run_until_update(max_env_steps=seen - 1)
    states, actions, rewards, terms = st.buffer()
    assert len(actions) == ts["n"] - 1
    taken, ts2, _ = st.run_until_update(max_env_steps=1)
    assert ts2["trained"] and ts2["n"] == ts["n"] and ts2["critic_loss"] == ts["critic_loss"]
    assert not np.array_equal(st.get_params(), params)
    # terminal rewards are 0 (Q12) and every terminal starts a fresh episode in the buffer
    assert all(rewards[terms == 1] == 0.0) and all(rewards[terms == 0] == 1.0)
    st.close()
