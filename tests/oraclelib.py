"""ctypes binding of oracle/libppo_oracle.so — the CPU parity oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/ppo_oracle.h). The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libppo_oracle.so")
if os.environ.get("CRL_ORACLE_SO"):   # e.g. an ASan/UBSan build (scripts/oracle_sanitize.sh)
    _SO = os.environ["CRL_ORACLE_SO"]


class OrcConfig(C.Structure):
    _fields_ = [
        ("num_steps", C.c_int32), ("num_envs", C.c_int32), ("num_minibatches", C.c_int32),
        ("update_epochs", C.c_int32), ("lr", C.c_float), ("gamma", C.c_float), ("gae_lambda", C.c_float),
        ("clip_coef", C.c_float), ("ent_coeff", C.c_float), ("v_coef", C.c_float),
        ("normalize_advantages", C.c_int32), ("clip_value_loss", C.c_int32), ("anneal_lr", C.c_int32),
        ("obs_dim", C.c_int32), ("n_act", C.c_int32), ("hidden", C.c_int32), ("gae_mode", C.c_int32),
        ("env_kind", C.c_int32), ("stale_obs", C.c_int32), ("env_id_offset", C.c_int32), ("seed", C.c_uint64),
    ]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("loss", "pg_loss", "v_loss", "entropy_loss", "adv_mean", "adv_std", "u", "n_unclipped_wins")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class OrcState(C.Structure):
    _fields_ = [
        ("obs", C.POINTER(C.c_float)), ("action", C.POINTER(C.c_int32)), ("logprob", C.POINTER(C.c_float)),
        ("reward", C.POINTER(C.c_float)), ("terminal", C.POINTER(C.c_uint8)), ("value", C.POINTER(C.c_float)),
        ("adv", C.POINTER(C.c_float)), ("ret", C.POINTER(C.c_float)),
        ("env_state", C.POINTER(C.c_float)), ("env_t", C.POINTER(C.c_int32)), ("cur_obs", C.POINTER(C.c_float)),
        ("next_done", C.POINTER(C.c_uint8)), ("ep_return", C.POINTER(C.c_float)), ("ep_length", C.POINTER(C.c_int32)),
        ("ep_count", C.c_double), ("ep_return_sum", C.c_double), ("ep_length_sum", C.c_double),
        ("params", C.POINTER(C.c_float)), ("grads", C.POINTER(C.c_float)), ("adam_m", C.POINTER(C.c_float)),
        ("adam_v", C.POINTER(C.c_float)), ("betap", C.c_double * 24), ("perm", C.POINTER(C.c_int32)),
        ("iteration", C.c_uint64),
    ]


def build(force=False):
    srcs = [os.path.join(_ROOT, "oracle", f) for f in ("ppo_oracle.c", "ppo_oracle.h", "a2c_oracle.c", "a2c_oracle.h", "dqn_oracle.c", "dqn_oracle.h")]
    if os.environ.get("CRL_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, ip, dp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        cp = C.POINTER(OrcConfig)
        L.orc_param_count.restype = C.c_int32
        L.orc_param_count.argtypes = [cp]
        L.orc_param_offsets.argtypes = [cp, ip]
        L.orc_tanh_fast.restype = C.c_float
        L.orc_tanh_fast.argtypes = [C.c_float]
        L.orc_sin_poly.restype = C.c_float
        L.orc_sin_poly.argtypes = [C.c_float]
        L.orc_cos_poly.restype = C.c_float
        L.orc_cos_poly.argtypes = [C.c_float]
        L.orc_u53.restype = C.c_double
        L.orc_u53.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32]
        L.orc_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.orc_mlp_forward.argtypes = [cp, fp, C.c_int, fp, fp, fp, fp]
        L.orc_get_action.argtypes = [cp, fp, fp, dp, C.c_int32, ip, fp, fp, dp]
        L.orc_logprob_actions.argtypes = [cp, fp, fp, ip, C.c_int32, fp, fp]
        L.orc_gae.argtypes = [fp, C.c_ssize_t, fp, C.c_ssize_t, u8p, C.c_ssize_t, C.c_int32, C.c_float, C.c_float,
                              C.c_int32, fp, C.c_ssize_t]
        L.orc_gae_batch.argtypes = [fp, fp, u8p, fp, u8p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, fp, fp]
        L.orc_loss_grad.argtypes = [cp, fp, fp, ip, fp, fp, fp, fp, ip, C.c_int32, dp, fp, C.POINTER(OrcStats)]
        L.orc_clipnorm_adam.argtypes = [cp, fp, fp, fp, fp, dp, C.c_double, C.c_double]
        L.orc_cartpole_step.argtypes = [fp, ip, C.c_int32, C.c_int32, ip]
        L.orc_env_reset.argtypes = [cp, fp, C.c_uint32, C.c_uint64, C.c_uint32]
        L.orc_state_create.restype = C.POINTER(OrcState)
        L.orc_state_create.argtypes = [cp]
        L.orc_state_destroy.argtypes = [C.POINTER(OrcState)]
        L.orc_env_init.argtypes = [cp, C.POINTER(OrcState)]
        L.orc_rollout.argtypes = [cp, C.POINTER(OrcState)]
        L.orc_compute_gae.argtypes = [cp, C.POINTER(OrcState)]
        L.orc_shuffle_fy.argtypes = [ip, C.c_int32, C.c_uint64, C.c_uint64]
        L.orc_shuffle_blocked_fy.argtypes = [ip, C.c_int32, C.c_uint64, C.c_uint64]
        L.orc_update_minibatch.argtypes = [cp, C.POINTER(OrcState), C.c_int32, C.c_double, C.POINTER(OrcStats)]
        L.orc_iterate.argtypes = [cp, C.POINTER(OrcState), C.c_int32, C.c_int32, C.POINTER(OrcStats)]
        L.orc_cartpole_step_libm.argtypes = [fp, ip, C.c_int32, C.c_int32, ip]
        L.orc_cartpole_step_batch.argtypes = [fp, ip, C.c_int32, C.c_int32, fp, u8p]
        _lib = L
    return _lib


_batched = None


def batched_lib():
    """oracle/libppo_cpu_batched.so: the batched, -O3 -march=native CPU iteration behind bench.py's cpu_baseline.batched (NOT a
    parity oracle). Always rebuilt on the box that runs it (`make -B batched`): -march=native code must not travel between hosts."""
    global _batched
    if _batched is None:
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "-s", "-B", "batched"])
        L = C.CDLL(os.path.join(_ROOT, "oracle", "libppo_cpu_batched.so"))
        cp = C.POINTER(OrcConfig)
        L.orc_state_create.restype = C.POINTER(OrcState); L.orc_state_create.argtypes = [cp]
        L.orc_state_destroy.argtypes = [C.POINTER(OrcState)]
        L.orc_env_init.argtypes = [cp, C.POINTER(OrcState)]
        L.orc_param_count.restype = C.c_int32; L.orc_param_count.argtypes = [cp]
        L.orc_batched_iterate.restype = C.c_int32
        L.orc_batched_iterate.argtypes = [cp, C.POINTER(OrcState), C.c_int32, C.c_int32, C.POINTER(OrcStats)]
        _batched = L
    return _batched


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def cartpole_step_batch(states, actions, libm):
    """One CartPole step from each of n given states (4, n): libm = True → sinf / cosf (the reference-side statement),
    False → the polynomial shared with the HIP env kernel. Returns (next states (4, n), done flags)."""
    st = np.asfortranarray(states, np.float32); n = st.shape[1]
    a = np.ascontiguousarray(actions, np.int32)
    out = np.zeros((4, n), np.float32, order="F"); dn = np.zeros(n, np.uint8)
    lib().orc_cartpole_step_batch(fptr(st), _p(a, C.c_int32), n, int(bool(libm)), fptr(out), _p(dn, C.c_uint8))
    return out, dn


def fptr(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous or a.flags.f_contiguous
    return _p(a, C.c_float)


def make_config(num_envs=8, num_steps=128, num_minibatches=4, update_epochs=4, lr=2.5e-4, gamma=0.99,
                gae_lambda=0.95, clip_coef=0.2, ent_coeff=0.01, v_coef=0.5, normalize_advantages=True,
                clip_value_loss=True, anneal_lr=True, obs_dim=4, n_act=2, hidden=64, gae_mode=0, env_kind=0,
                stale_obs=1, env_id_offset=0, seed=0x5EED):
    return OrcConfig(num_steps, num_envs, num_minibatches, update_epochs, lr, gamma, gae_lambda, clip_coef, ent_coeff,
                     v_coef, int(normalize_advantages), int(clip_value_loss), int(anneal_lr), obs_dim, n_act, hidden,
                     gae_mode, env_kind, stale_obs, env_id_offset, seed)


def param_offsets(cfg):
    o = np.zeros(13, np.int32)
    lib().orc_param_offsets(C.byref(cfg), _p(o, C.c_int32))
    return o


def get_action(cfg, params, obs, u, with_value=True):
    n = obs.shape[-1] if obs.ndim == 2 else 1
    obs = np.asfortranarray(obs, np.float32)
    action = np.zeros(n, np.int32); logprob = np.zeros(n, np.float32); value = np.zeros(n, np.float32)
    margin = np.zeros(n, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    lib().orc_get_action(C.byref(cfg), fptr(params), fptr(obs), _p(u, C.c_double), n, _p(action, C.c_int32),
                         fptr(logprob), fptr(value) if with_value else None, _p(margin, C.c_double))
    return action, logprob, value, margin


def logprob_actions(cfg, params, obs, actions):
    n = obs.shape[-1]
    obs = np.asfortranarray(obs, np.float32)
    actions = np.ascontiguousarray(actions, np.int32)
    logprob = np.zeros(n, np.float32); ent = np.zeros((cfg.n_act, n), np.float32, order="F")
    lib().orc_logprob_actions(C.byref(cfg), fptr(params), fptr(obs), _p(actions, C.c_int32), n, fptr(logprob), fptr(ent))
    return logprob, ent


def gae(values, rewards, terminals, gamma, lam, mode=0):
    values = np.ascontiguousarray(values, np.float32); rewards = np.ascontiguousarray(rewards, np.float32)
    terminals = np.ascontiguousarray(terminals, np.uint8)
    k = rewards.shape[0]
    adv = np.zeros(k, np.float32)
    lib().orc_gae(fptr(values), 1, fptr(rewards), 1, _p(terminals, C.c_uint8), 1, k, gamma, lam, mode, fptr(adv), 1)
    return adv


def gae_batch(value, reward, terminal, next_value, next_done, gamma, lam, mode=0):
    """value/reward/terminal are (nt,k) Fortran-ordered."""
    nt, k = value.shape
    value = np.asfortranarray(value, np.float32); reward = np.asfortranarray(reward, np.float32)
    terminal = np.asfortranarray(terminal, np.uint8)
    next_value = np.ascontiguousarray(next_value, np.float32); next_done = np.ascontiguousarray(next_done, np.uint8)
    adv = np.zeros((nt, k), np.float32, order="F"); ret = np.zeros((nt, k), np.float32, order="F")
    lib().orc_gae_batch(fptr(value), fptr(reward), _p(terminal, C.c_uint8), fptr(next_value), _p(next_done, C.c_uint8),
                        nt, k, gamma, lam, mode, fptr(adv), fptr(ret))
    return adv, ret


def loss_grad(cfg, params, states, actions, logprobs, values, advantages, returns, mb_inds, adv_stats=None):
    P = lib().orc_param_count(C.byref(cfg))
    grads = np.zeros(P, np.float32)
    st = OrcStats()
    states = np.asfortranarray(states, np.float32)
    arrs = [np.ascontiguousarray(a, np.float32).ravel(order="F") for a in (logprobs, values, advantages, returns)]
    actions = np.ascontiguousarray(actions, np.int32).ravel(order="F")
    mb = np.ascontiguousarray(mb_inds, np.int32)
    stats = None
    if adv_stats is not None:
        stats = np.ascontiguousarray(adv_stats, np.float64)
    lib().orc_loss_grad(C.byref(cfg), fptr(params), fptr(states), _p(actions, C.c_int32), fptr(arrs[0]), fptr(arrs[1]),
                        fptr(arrs[2]), fptr(arrs[3]), _p(mb, C.c_int32), len(mb),
                        _p(stats, C.c_double) if stats is not None else None, fptr(grads), C.byref(st))
    return grads, st.as_dict()


def clipnorm_adam(cfg, params, grads, m, v, betap, eta, thresh=0.5):
    lib().orc_clipnorm_adam(C.byref(cfg), fptr(params), fptr(grads), fptr(m), fptr(v), _p(betap, C.c_double), eta, thresh)


class State:
    """Owns an orc_state and exposes its arrays as numpy views."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.ptr = lib().orc_state_create(C.byref(cfg))
        s = self.ptr.contents
        nt, k, d = cfg.num_envs, cfg.num_steps, cfg.obs_dim
        B = nt * k
        self.P = lib().orc_param_count(C.byref(cfg))
        v = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).view(dt) if n else None
        self.obs = np.ctypeslib.as_array(s.obs, shape=(B * d,)).reshape((d, nt, k), order="F")
        self.action = np.ctypeslib.as_array(s.action, shape=(B,)).reshape((nt, k), order="F")
        self.logprob = np.ctypeslib.as_array(s.logprob, shape=(B,)).reshape((nt, k), order="F")
        self.reward = np.ctypeslib.as_array(s.reward, shape=(B,)).reshape((nt, k), order="F")
        self.terminal = np.ctypeslib.as_array(s.terminal, shape=(B,)).reshape((nt, k), order="F")
        self.value = np.ctypeslib.as_array(s.value, shape=(B,)).reshape((nt, k), order="F")
        self.adv = np.ctypeslib.as_array(s.adv, shape=(B,)).reshape((nt, k), order="F")
        self.ret = np.ctypeslib.as_array(s.ret, shape=(B,)).reshape((nt, k), order="F")
        self.env_state = np.ctypeslib.as_array(s.env_state, shape=(nt * d,)).reshape((d, nt), order="F")
        self.env_t = np.ctypeslib.as_array(s.env_t, shape=(nt,))
        self.cur_obs = np.ctypeslib.as_array(s.cur_obs, shape=(nt * d,)).reshape((d, nt), order="F")
        self.next_done = np.ctypeslib.as_array(s.next_done, shape=(nt,))
        self.ep_return = np.ctypeslib.as_array(s.ep_return, shape=(nt,))
        self.ep_length = np.ctypeslib.as_array(s.ep_length, shape=(nt,))
        self.params = np.ctypeslib.as_array(s.params, shape=(self.P,))
        self.grads = np.ctypeslib.as_array(s.grads, shape=(self.P,))
        self.adam_m = np.ctypeslib.as_array(s.adam_m, shape=(self.P,))
        self.adam_v = np.ctypeslib.as_array(s.adam_v, shape=(self.P,))
        self.perm = np.ctypeslib.as_array(s.perm, shape=(B,))
        del v

    @property
    def iteration(self):
        return self.ptr.contents.iteration

    @property
    def episode_stats(self):
        s = self.ptr.contents
        return s.ep_count, s.ep_return_sum, s.ep_length_sum

    @property
    def betap(self):
        return np.array(list(self.ptr.contents.betap))

    def env_init(self):
        lib().orc_env_init(C.byref(self.cfg), self.ptr)

    def rollout(self):
        lib().orc_rollout(C.byref(self.cfg), self.ptr)

    def compute_gae(self):
        lib().orc_compute_gae(C.byref(self.cfg), self.ptr)

    def update_minibatch(self, mb, eta):
        st = OrcStats()
        lib().orc_update_minibatch(C.byref(self.cfg), self.ptr, mb, eta, C.byref(st))
        return st.as_dict()

    def iterate(self, num_updates_total, gen_perm=True):
        n = self.cfg.update_epochs * self.cfg.num_minibatches
        arr = (OrcStats * n)()
        lib().orc_iterate(C.byref(self.cfg), self.ptr, num_updates_total, int(gen_perm), arr)
        return [a.as_dict() for a in arr]

    def batched_iterate(self, num_updates_total, gen_perm=True):
        """The same loop body through oracle/ppo_cpu_batched.c (throughput baseline; the state layout is shared)."""
        n = self.cfg.update_epochs * self.cfg.num_minibatches
        arr = (OrcStats * n)()
        rc = batched_lib().orc_batched_iterate(C.byref(self.cfg), self.ptr, num_updates_total, int(gen_perm), arr)
        if rc:
            raise ValueError("orc_batched_iterate covers hidden 64 / CartPole / compat GAE only")
        return [a.as_dict() for a in arr]

    def close(self):
        if self.ptr:
            lib().orc_state_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shuffle_fy(perm, seed, epoch_id):
    perm = np.ascontiguousarray(perm, np.int32)
    lib().orc_shuffle_fy(_p(perm, C.c_int32), len(perm), seed, epoch_id)
    return perm


def shuffle_blocked_fy(n, seed, epoch_id):
    perm = np.zeros(n, np.int32)
    lib().orc_shuffle_blocked_fy(_p(perm, C.c_int32), n, seed, epoch_id)
    return perm


def orthogonal_params(cfg, seed=0):
    """Flux.orthogonal-shaped init (networks.jl:9,40-41): gains sqrt(2) hidden, 0.01 actor head, 1.0 critic head,
    zero biases. QR of a Gaussian with sign fix — Flux's stream is not reproducible here; weights are an INPUT."""
    rng = np.random.default_rng(seed)
    o = param_offsets(cfg)
    P = int(o[12])
    out = np.zeros(P, np.float32)
    h, d, A = cfg.hidden, cfg.obs_dim, cfg.n_act

    def orth(rows, cols, gain):
        a = rng.standard_normal((max(rows, cols), min(rows, cols)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if rows < cols:
            q = q.T
        return (gain * q[:rows, :cols]).astype(np.float32)

    shapes = [(h, d, np.sqrt(2)), None, (h, h, np.sqrt(2)), None, (A, h, 0.01), None,
              (h, d, np.sqrt(2)), None, (h, h, np.sqrt(2)), None, (1, h, 1.0), None]
    for i, sh in enumerate(shapes):
        if sh is None:
            continue
        W = orth(sh[0], sh[1], sh[2])
        out[o[i]:o[i + 1]] = W.ravel(order="F")
    return out


# ---------------------------------------------------------------------------------------------------------
# A2C oracle (oracle/a2c_oracle.c — a2c.jl restated)
# ---------------------------------------------------------------------------------------------------------
class A2CConfigC(C.Structure):
    _fields_ = [("lr", C.c_double), ("total_timesteps", C.c_int64), ("min_replay_size", C.c_int32), ("gamma", C.c_double),
                ("obs_dim", C.c_int32), ("n_act", C.c_int32), ("hidden", C.c_int32), ("max_steps", C.c_int32),
                ("seed", C.c_uint64)]


class A2CTrainStats(C.Structure):
    _fields_ = [("actor_loss", C.c_double), ("critic_loss", C.c_double), ("n", C.c_int32), ("trained", C.c_int32)]


class A2CEpisode(C.Structure):
    _fields_ = [("episode_return", C.c_double), ("episode_length", C.c_int64), ("global_step", C.c_int64)]


_a2c_ready = False


def a2c_lib():
    global _a2c_ready
    L = lib()
    if not _a2c_ready:
        dp, fp, ip, u8p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        cp = C.POINTER(A2CConfigC)
        L.a2c_create.restype = C.c_void_p; L.a2c_create.argtypes = [cp]
        L.a2c_destroy.argtypes = [C.c_void_p]
        L.a2c_param_count.restype = C.c_int32; L.a2c_param_count.argtypes = [C.c_void_p]
        L.a2c_set_params.argtypes = [C.c_void_p, fp]; L.a2c_get_params.argtypes = [C.c_void_p, fp]
        L.a2c_get_env.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64), ip]
        L.a2c_get_buffer.argtypes = [C.c_void_p, dp, ip, dp, u8p]
        L.a2c_run_until_update.restype = C.c_int64
        L.a2c_run_until_update.argtypes = [C.c_void_p, C.c_int64, C.POINTER(A2CTrainStats), C.POINTER(A2CEpisode), C.c_int32, ip]
        L.a2c_discounted_future_rewards.argtypes = [dp, u8p, C.c_int32, C.c_double, C.c_double, dp]
        for f in (L.a2c_tanh_fast, L.a2c_sin, L.a2c_cos):
            f.restype = C.c_double; f.argtypes = [C.c_double]
        L.a2c_forward.argtypes = [cp, fp, C.c_int, dp, dp]
        L.a2c_loss_grads.argtypes = [cp, fp, dp, ip, dp, C.c_int32, fp, dp, dp, dp]
        L.a2c_cartpole_step.argtypes = [dp, ip, C.c_int32, C.c_int32, ip]
        _a2c_ready = True
    return L


def a2c_config(lr=1e-4, total_timesteps=1_000_000, min_replay_size=512, gamma=0.99, obs_dim=4, n_act=2, hidden=64,
               max_steps=500, seed=0x5EED):
    return A2CConfigC(lr, total_timesteps, min_replay_size, gamma, obs_dim, n_act, hidden, max_steps, seed)


def a2c_discounted_future_rewards(rewards, terminals, final_value, gamma):
    r = np.ascontiguousarray(rewards, np.float64); t = np.ascontiguousarray(terminals, np.uint8)
    out = np.zeros(len(r), np.float64)
    a2c_lib().a2c_discounted_future_rewards(_p(r, C.c_double), _p(t, C.c_uint8), len(r), final_value, gamma, _p(out, C.c_double))
    return out


def a2c_forward(cfg, params, net, x):
    params = np.ascontiguousarray(params, np.float32); x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(1 if net else cfg.n_act, np.float64)
    a2c_lib().a2c_forward(C.byref(cfg), _p(params, C.c_float), net, _p(x, C.c_double), _p(out, C.c_double))
    return out


def a2c_loss_grads(cfg, params, states, actions, returns):
    params = np.ascontiguousarray(params, np.float32)
    states = np.asfortranarray(states, np.float64); actions = np.ascontiguousarray(actions, np.int32)
    returns = np.ascontiguousarray(returns, np.float64)
    n = len(actions)
    grads = np.zeros(len(params), np.float32); adv = np.zeros(n, np.float64)
    cl, al = C.c_double(), C.c_double()
    a2c_lib().a2c_loss_grads(C.byref(cfg), _p(params, C.c_float), states.ctypes.data_as(C.POINTER(C.c_double)), _p(actions, C.c_int32),
                             _p(returns, C.c_double), n, _p(grads, C.c_float), C.byref(cl), C.byref(al), _p(adv, C.c_double))
    return grads, cl.value, al.value, adv


class A2CState:
    def __init__(self, cfg, params):
        self.cfg = cfg
        self.L = a2c_lib()
        self.ptr = self.L.a2c_create(C.byref(cfg))
        self.P = self.L.a2c_param_count(self.ptr)
        self.set_params(params)

    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32); assert len(p) == self.P
        self.L.a2c_set_params(self.ptr, _p(p, C.c_float))

    def get_params(self):
        p = np.zeros(self.P, np.float32); self.L.a2c_get_params(self.ptr, _p(p, C.c_float)); return p

    def env(self):
        s = np.zeros(4, np.float64); g = C.c_int64(); n = C.c_int32()
        self.L.a2c_get_env(self.ptr, _p(s, C.c_double), C.byref(g), C.byref(n))
        return s, g.value, n.value

    def buffer(self):
        _, _, n = self.env()
        d = self.cfg.obs_dim
        st = np.zeros((d, n), np.float64, order="F"); a = np.zeros(n, np.int32); r = np.zeros(n, np.float64); t = np.zeros(n, np.uint8)
        self.L.a2c_get_buffer(self.ptr, st.ctypes.data_as(C.POINTER(C.c_double)), _p(a, C.c_int32), _p(r, C.c_double), _p(t, C.c_uint8))
        return st, a, r, t

    def run_until_update(self, max_env_steps=1 << 40, max_eps=4096):
        ts = A2CTrainStats(); eps = (A2CEpisode * max_eps)(); n = C.c_int32()
        taken = self.L.a2c_run_until_update(self.ptr, max_env_steps, C.byref(ts), eps, max_eps, C.byref(n))
        episodes = [(eps[i].episode_return, eps[i].episode_length, eps[i].global_step) for i in range(n.value)]
        return taken, dict(actor_loss=ts.actor_loss, critic_loss=ts.critic_loss, n=ts.n, trained=bool(ts.trained)), episodes

    def close(self):
        if self.ptr:
            self.L.a2c_destroy(self.ptr); self.ptr = None


# ---------------------------------------------------------------------------------------------------------
# DQN oracle (oracle/dqn_oracle.c — dqn.jl restated)
# ---------------------------------------------------------------------------------------------------------
DQN_P = 120 * 4 + 120 + 84 * 120 + 84 + 2 * 84 + 2
DQN_OFF = np.cumsum([0, 480, 120, 10080, 84, 168, 2])


class DQNConfigC(C.Structure):
    _fields_ = [("log_frequency", C.c_int64), ("total_timesteps", C.c_int64), ("buffer_size", C.c_int64), ("min_buff_size", C.c_int64),
                ("lr", C.c_double), ("train_freq", C.c_int64), ("target_net_freq", C.c_int64), ("batch_size", C.c_int64),
                ("gamma", C.c_double), ("epsilon_start", C.c_double), ("epsilon_end", C.c_double), ("epsilon_duration", C.c_double),
                ("max_steps", C.c_int32), ("pad", C.c_int32), ("seed", C.c_uint64)]


class DQNEpisode(C.Structure):
    _fields_ = [("episode_return", C.c_double), ("episode_length", C.c_int64), ("global_step", C.c_int64), ("epsilon", C.c_double)]


class DQNLossRecord(C.Structure):
    _fields_ = [("global_step", C.c_int64), ("loss", C.c_double)]


_dqn_ready = False


def dqn_lib():
    global _dqn_ready
    L = lib()
    if not _dqn_ready:
        dp, fp, ip, u8p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_int64)
        cp = C.POINTER(DQNConfigC)
        L.dqn_create.restype = C.c_void_p; L.dqn_create.argtypes = [cp]
        L.dqn_destroy.argtypes = [C.c_void_p]
        L.dqn_set_params.argtypes = [C.c_void_p, fp]; L.dqn_get_params.argtypes = [C.c_void_p, fp, fp]
        L.dqn_get_env.argtypes = [C.c_void_p, dp, i64p, i64p, dp, i64p]
        L.dqn_run.restype = C.c_int64
        L.dqn_run.argtypes = [C.c_void_p, C.c_int64, C.POINTER(DQNEpisode), C.c_int32, ip, C.POINTER(DQNLossRecord), C.c_int32, ip]
        L.dqn_linear_schedule.restype = C.c_double; L.dqn_linear_schedule.argtypes = [C.c_double] * 4
        L.dqn_forward.argtypes = [fp, dp, dp]
        L.dqn_loss_grads.restype = C.c_double
        L.dqn_loss_grads.argtypes = [fp, fp, dp, dp, ip, dp, u8p, C.c_int32, C.c_double, fp]
        L.dqn_sample_indices.argtypes = [C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, ip]
        _dqn_ready = True
    return L


def dqn_config(log_frequency=1000, total_timesteps=500_000, buffer_size=10_000, min_buff_size=200, lr=1e-4, train_freq=10,
               target_net_freq=100, batch_size=120, gamma=0.99, epsilon_start=1.0, epsilon_end=0.05, epsilon_duration=10_000.0,
               max_steps=200, seed=0x5EED):
    return DQNConfigC(log_frequency, total_timesteps, buffer_size, min_buff_size, lr, train_freq, target_net_freq, batch_size, gamma,
                      epsilon_start, epsilon_end, epsilon_duration, max_steps, 0, seed)


def dqn_params(seed=0):
    """Flux Dense default init [3P-memory]: glorot_uniform Float32 weights, zero biases. Flux's RNG stream is not
    reproducible here — weights are an INPUT of both sides."""
    rng = np.random.default_rng(seed)
    out = np.zeros(DQN_P, np.float32)
    for i, (rows, cols) in zip((0, 2, 4), ((120, 4), (84, 120), (2, 84))):
        lim = np.sqrt(6.0 / (rows + cols))
        out[DQN_OFF[i]:DQN_OFF[i + 1]] = rng.uniform(-lim, lim, rows * cols).astype(np.float32)
    return out


def dqn_forward(params, x):
    params = np.ascontiguousarray(params, np.float32); x = np.ascontiguousarray(x, np.float64)
    q = np.zeros(2, np.float64)
    dqn_lib().dqn_forward(_p(params, C.c_float), _p(x, C.c_double), _p(q, C.c_double))
    return q


def dqn_loss_grads(q_params, t_params, state, next_state, action, reward, terminal, gamma):
    qp = np.ascontiguousarray(q_params, np.float32); tp = np.ascontiguousarray(t_params, np.float32)
    st = np.asfortranarray(state, np.float64); nx = np.asfortranarray(next_state, np.float64)
    a = np.ascontiguousarray(action, np.int32); r = np.ascontiguousarray(reward, np.float64); t = np.ascontiguousarray(terminal, np.uint8)
    g = np.zeros(DQN_P, np.float32)
    loss = dqn_lib().dqn_loss_grads(_p(qp, C.c_float), _p(tp, C.c_float), st.ctypes.data_as(C.POINTER(C.c_double)),
                                    nx.ctypes.data_as(C.POINTER(C.c_double)), _p(a, C.c_int32), _p(r, C.c_double), _p(t, C.c_uint8),
                                    len(a), gamma, _p(g, C.c_float))
    return loss, g


def dqn_sample_indices(seed, gstep, n, k):
    out = np.zeros(k, np.int32)
    dqn_lib().dqn_sample_indices(seed, gstep, n, k, _p(out, C.c_int32))
    return out


class DQNState:
    def __init__(self, cfg, params):
        self.cfg = cfg; self.L = dqn_lib()
        self.ptr = self.L.dqn_create(C.byref(cfg))
        p = np.ascontiguousarray(params, np.float32); assert p.size == DQN_P
        self.L.dqn_set_params(self.ptr, _p(p, C.c_float))

    def params(self):
        q = np.zeros(DQN_P, np.float32); t = np.zeros(DQN_P, np.float32)
        self.L.dqn_get_params(self.ptr, _p(q, C.c_float), _p(t, C.c_float))
        return q, t

    def env(self):
        s = np.zeros(4, np.float64); g = C.c_int64(); n = C.c_int64(); ll = C.c_double(); nu = C.c_int64()
        self.L.dqn_get_env(self.ptr, _p(s, C.c_double), C.byref(g), C.byref(n), C.byref(ll), C.byref(nu))
        return dict(state=s, global_step=g.value, rb_size=n.value, last_loss=ll.value, n_updates=nu.value)

    def run(self, max_env_steps, max_eps=8192, max_losses=4096):
        eps = (DQNEpisode * max_eps)(); ls = (DQNLossRecord * max_losses)(); ne = C.c_int32(); nl = C.c_int32()
        taken = self.L.dqn_run(self.ptr, max_env_steps, eps, max_eps, C.byref(ne), ls, max_losses, C.byref(nl))
        return (taken, [(eps[i].episode_return, eps[i].episode_length, eps[i].global_step, eps[i].epsilon) for i in range(ne.value)],
                [(ls[i].global_step, ls[i].loss) for i in range(nl.value)])

    def close(self):
        if self.ptr:
            self.L.dqn_destroy(self.ptr); self.ptr = None
