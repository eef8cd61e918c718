"""Pins for the DQN oracle (oracle/dqn_oracle.c, restating dqn.jl): hand-derived known answers for linear_schedule
(dqn.jl:28-31), an independent torch-float64 autograd restatement of the TD-target / mse closure (dqn.jl:96-108), sampler
properties, and loop-level invariants of dqn.jl:57-119."""
import numpy as np
import torch

import oraclelib as O


def test_linear_schedule_known_answers():
    L = O.dqn_lib()
    assert L.dqn_linear_schedule(1.0, 0.05, 10_000.0, 0.0) == 1.0
    assert abs(L.dqn_linear_schedule(1.0, 0.05, 10_000.0, 5000.0) - 0.525) < 1e-15
    assert L.dqn_linear_schedule(1.0, 0.05, 10_000.0, 10_000.0) == max((0.05 - 1.0) / 10_000.0 * 10_000.0 + 1.0, 0.05)
    assert L.dqn_linear_schedule(1.0, 0.05, 10_000.0, 1e6) == 0.05


def _torch_q(p, x):
    o = O.DQN_OFF
    W1 = p[o[0]:o[1]].reshape(4, 120).T; b1 = p[o[1]:o[2]]
    W2 = p[o[2]:o[3]].reshape(120, 84).T; b2 = p[o[3]:o[4]]
    W3 = p[o[4]:o[5]].reshape(84, 2).T; b3 = p[o[5]:o[6]]
    h1 = torch.relu(W1 @ x + b1[:, None]); h2 = torch.relu(W2 @ h1 + b2[:, None])
    return W3 @ h2 + b3[:, None]


def test_loss_grads_match_torch_autograd():
    rng = np.random.default_rng(2)
    qp = O.dqn_params(1) + (0.02 * rng.standard_normal(O.DQN_P)).astype(np.float32)
    tp = O.dqn_params(1) + (0.02 * rng.standard_normal(O.DQN_P)).astype(np.float32)
    n = 120
    state = rng.standard_normal((4, n)); nxt = rng.standard_normal((4, n))
    act = rng.integers(0, 2, n).astype(np.int32); rew = (rng.random(n) > 0.1).astype(np.float64); term = (rng.random(n) < 0.2).astype(np.uint8)
    loss, g = O.dqn_loss_grads(qp, tp, state, nxt, act, rew, term, 0.99)
    p64 = torch.tensor(qp.astype(np.float64), requires_grad=True)
    with torch.no_grad():
        next_q = _torch_q(torch.tensor(tp.astype(np.float64)), torch.tensor(nxt)).max(0).values           # dqn.jl:99
    td = torch.tensor(rew) + 0.99 * next_q * (1.0 - torch.tensor(term.astype(np.float64)))                   # dqn.jl:100
    q = _torch_q(p64, torch.tensor(state))[torch.tensor(act.astype(np.int64)), torch.arange(n)]              # dqn.jl:105-106
    tl = ((td - q) ** 2).mean()                                                                             # Flux.mse
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12 * max(1, abs(loss))
    gt = p64.grad.numpy()
    for i in range(6):
        a, b = g[O.DQN_OFF[i]:O.DQN_OFF[i + 1]].astype(np.float64), gt[O.DQN_OFF[i]:O.DQN_OFF[i + 1]]
        assert np.linalg.norm(a - b) <= 1e-6 * max(np.linalg.norm(b), 1e-12), i
    # q values themselves
    assert np.allclose(O.dqn_forward(qp, state[:, 3]), _torch_q(torch.tensor(qp.astype(np.float64)), torch.tensor(state[:, 3:4]))[:, 0].numpy(), atol=1e-13)


def test_sampler_draws_distinct_uniform_indices():
    for n, k in ((120, 120), (210, 120), (10_000, 120), (5, 1)):
        idx = O.dqn_sample_indices(7, 1234, n, k)
        assert len(set(idx.tolist())) == k and idx.min() >= 0 and idx.max() < n
    assert not np.array_equal(O.dqn_sample_indices(7, 10, 1000, 120), O.dqn_sample_indices(7, 20, 1000, 120))
    # uniformity over many steps: every index of a 300-slot buffer is hit ≈ k/n of the time
    hits = np.zeros(300)
    for g in range(2000):
        hits[O.dqn_sample_indices(1, g, 300, 120)] += 1
    assert abs(hits.mean() / 2000 - 0.4) < 1e-12 and hits.std() / 2000 < 0.02
    # first position is uniform too (a k-PERMUTATION, not just a subset)
    first = np.array([O.dqn_sample_indices(1, g, 300, 120)[0] for g in range(3000)])
    assert abs(first.mean() - 149.5) < 8


def test_loop_invariants():
    cfg = O.dqn_config(total_timesteps=1500, seed=5)
    st = O.DQNState(cfg, O.dqn_params(3))
    taken, eps, losses = st.run(205)
    e = st.env()
    assert taken == 205 and e["n_updates"] == 0 and e["rb_size"] == 205            # dqn.jl:93: first update at step 210
    taken, eps2, losses = st.run(5)
    assert st.env()["n_updates"] == 1 and st.env()["last_loss"] > 0
    q1, t1 = st.params()
    assert not np.array_equal(q1, t1), "target net only follows every target_net_freq steps"
    taken, eps3, losses = st.run(90)                                                # … step 300: a multiple of 100
    q2, t2 = st.params()
    assert np.array_equal(q2, t2) and st.env()["n_updates"] == 10
    taken, eps4, losses = st.run(10_000)
    assert st.env()["global_step"] == 1500 and taken == 1200
    assert [g for g, _ in losses] == [1000]                                          # log_frequency (dqn.jl:115)
    for ret, length, gstep, eps_t in eps + eps2 + eps3 + eps4:
        assert ret == length - 1 or length == 201, "reward 0 on the terminating step; time-limit episodes run 201 steps"
        assert abs(eps_t - max((0.05 - 1.0) / 10_000.0 * gstep + 1.0, 0.05)) < 1e-15
    st.close()
