"""Networks.make_actor_critic (networks.jl:36-53): shapes, orthogonal gains and zero biases of the reference's
separate actor / critic MLPs, returned as ONE flat float32 vector in Flux.params(actor, critic) order (ppo.jl:196).
Host-side initialisation only (run once); Flux's own RNG stream is not reproducible, weights are an input."""
import numpy as np


def param_offsets(n_act, obs_dim, hidden_sizes=(64, 64)):
    h1, h2 = hidden_sizes
    if h1 != h2:
        raise ValueError("equal hidden sizes only")
    sizes = [h1 * obs_dim, h1, h2 * h1, h2, n_act * h2, n_act, h1 * obs_dim, h1, h2 * h1, h2, h2, 1]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def _orthogonal(rng, rows, cols, gain):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


def make_actor_critic(n_act, obs_dim, hidden_sizes=(64, 64), seed=0):
    """Dense(in,64,tanh_fast) → Dense(64,64,tanh_fast) → Dense(64,out); gains √2 / 0.01 (actor head) / 1.0 (critic
    head) (networks.jl:6-13,40-46)."""
    rng = np.random.default_rng(seed)
    off = param_offsets(n_act, obs_dim, hidden_sizes)
    h = hidden_sizes[0]
    out = np.zeros(int(off[-1]), np.float32)
    shapes = {0: (h, obs_dim, np.sqrt(2)), 2: (h, h, np.sqrt(2)), 4: (n_act, h, 0.01),
              6: (h, obs_dim, np.sqrt(2)), 8: (h, h, np.sqrt(2)), 10: (1, h, 1.0)}
    for i, (r, c, g) in shapes.items():
        out[off[i]:off[i + 1]] = _orthogonal(rng, r, c, g).ravel(order="F")
    return out
