"""ctypes binding of libcleanrl_hip.so (include/cleanrl_hip.h).

There is NO CPU fallback anywhere in this package: if the HIP library is missing the import of this module raises,
and every compute entry point returns an error when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRL_LIB_PATH") or os.path.join(_HERE, "libcleanrl_hip.so")   # CRL_LIB_PATH: build-variant experiments


class CrlError(RuntimeError):
    pass


class CrlConfig(C.Structure):
    """crl_ppo_config — mirror of PPOConfig (ppo.jl:1-19) + shapes."""
    _fields_ = [
        ("total_timesteps", C.c_int64), ("num_steps", C.c_int32), ("num_envs", C.c_int32),
        ("num_minibatches", C.c_int32), ("update_epochs", C.c_int32), ("lr", C.c_float), ("gamma", C.c_float),
        ("gae_lambda", C.c_float), ("clip_coef", C.c_float), ("ent_coeff", C.c_float), ("v_coef", C.c_float),
        ("normalize_advantages", C.c_int32), ("clip_value_loss", C.c_int32), ("anneal_lr", C.c_int32),
        ("obs_dim", C.c_int32), ("n_act", C.c_int32), ("hidden", C.c_int32), ("gae_mode", C.c_int32),
        ("env_kind", C.c_int32), ("stale_obs", C.c_int32), ("env_id_offset", C.c_int32), ("shuffle_mode", C.c_int32),
        ("seed", C.c_uint64),
    ]


class CrlStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("loss", "pg_loss", "v_loss", "entropy_loss", "adv_mean", "adv_std", "u_value", "n_unclipped_wins")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CrlEpisodeRecord(C.Structure):
    _fields_ = [("episode_return", C.c_float), ("episode_length", C.c_int32), ("env", C.c_int32), ("step", C.c_int32)]


class CrlEpisodeStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("episodes", "return_sum", "length_sum", "return_max")]


class CrlIterationReport(C.Structure):
    """crl_ppo_iteration_report: whose records crl_ppo_iterate_async / crl_ppo_drain handed back."""
    _fields_ = [("iteration", C.c_int64), ("episodes", CrlEpisodeStats), ("n_episodes", C.c_int64), ("n_ring", C.c_int32), ("pad", C.c_int32)]


# every symbol include/cleanrl_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "crl_version", "crl_last_error", "crl_device_count", "crl_ppo_create", "crl_ppo_destroy", "crl_ppo_param_count",
    "crl_sync", "crl_ppo_write", "crl_ppo_read", "crl_policy_act", "crl_logprob_actions", "crl_gae",
    "crl_rollout_store", "crl_env_reset", "crl_rollout_run", "crl_episode_stats_read", "crl_compute_gae",
    "crl_shuffle", "crl_adv_stats", "crl_ppo_update_minibatch", "crl_ppo_iterate", "crl_ppo_iteration",
    "crl_comm_unique_id", "crl_comm_init", "crl_comm_init_external", "crl_comm_peer_export", "crl_comm_peer_attach", "crl_adv_stats_local", "crl_adv_stats_finish",
    "crl_prof_enable", "crl_prof_read", "crl_prof_reset", "crl_ppo_exact_reruns", "crl_episode_ring_enable",
    "crl_episode_ring_read", "crl_comm_destroy", "crl_ppo_set_option", "crl_ppo_get_option", "crl_ppo_option_name", "crl_ppo_option_count", "crl_gae_bench", "crl_gae_opt",
    "crl_a2c_create", "crl_a2c_destroy", "crl_a2c_param_count", "crl_a2c_write_params", "crl_a2c_read_params",
    "crl_a2c_read_env", "crl_a2c_read_buffer", "crl_a2c_run_until_update", "crl_a2c_discounted_future_rewards",
    "crl_dqn_create", "crl_dqn_destroy", "crl_dqn_write_params", "crl_dqn_read_params", "crl_dqn_status_read", "crl_dqn_run",
    "crl_dqn_q_values",
    "crl_make_actor_critic", "crl_ppo_init_params", "crl_a2c_init_params", "crl_dqn_make_nn", "crl_dqn_init_params", "crl_comm_info", "crl_clock_probe", "crl_product_probe", "crl_ppo_iterate_async", "crl_ppo_drain",
]

DQN_PARAM_COUNT = 10934


class CrlDQNConfig(C.Structure):
    """crl_dqn_config — mirror of DQNConfig (dqn.jl:1-19)."""
    _fields_ = [("log_frequency", C.c_int64), ("total_timesteps", C.c_int64), ("buffer_size", C.c_int64), ("min_buff_size", C.c_int64),
                ("lr", C.c_double), ("train_freq", C.c_int64), ("target_net_freq", C.c_int64), ("batch_size", C.c_int64),
                ("gamma", C.c_double), ("epsilon_start", C.c_double), ("epsilon_end", C.c_double), ("epsilon_duration", C.c_double),
                ("max_steps", C.c_int32), ("pad", C.c_int32), ("seed", C.c_uint64)]


class CrlDQNEpisode(C.Structure):
    _fields_ = [("episode_return", C.c_double), ("episode_length", C.c_int64), ("global_step", C.c_int64), ("epsilon", C.c_double)]


class CrlDQNLossRecord(C.Structure):
    _fields_ = [("global_step", C.c_int64), ("loss", C.c_double)]


class CrlDQNStatus(C.Structure):
    _fields_ = [("env_state", C.c_double * 4), ("global_step", C.c_int64), ("rb_size", C.c_int64), ("n_updates", C.c_int64),
                ("last_loss", C.c_double)]



class CrlA2CConfig(C.Structure):
    """crl_a2c_config — mirror of A2CConfig (a2c.jl:1-10)."""
    _fields_ = [("lr", C.c_double), ("total_timesteps", C.c_int64), ("min_replay_size", C.c_int32), ("max_steps", C.c_int32),
                ("gamma", C.c_double), ("seed", C.c_uint64)]


class CrlA2CTrainStats(C.Structure):
    _fields_ = [("actor_loss", C.c_double), ("critic_loss", C.c_double), ("n", C.c_int32), ("trained", C.c_int32)]


class CrlA2CEpisode(C.Structure):
    _fields_ = [("episode_return", C.c_double), ("episode_length", C.c_int64), ("global_step", C.c_int64)]


# crl_field
F_OBS, F_ACTION, F_LOGPROB, F_REWARD, F_TERMINAL, F_VALUE, F_ADVANTAGE, F_RETURN, F_PERM, F_PARAMS, F_GRADS, F_ADAM_M, \
    F_ADAM_V, F_ENV_STATE, F_CUR_OBS, F_NEXT_DONE, F_ENV_T, F_BETAP, F_ADV_SUMS = range(19)
GAE_COMPAT, GAE_FIXED = 0, 1
ENV_CARTPOLE, ENV_SYNTHETIC, ENV_EXTERNAL = 0, 1, 2
SHUFFLE_FISHER_YATES, SHUFFLE_BIJECTION, SHUFFLE_BLOCKED_FY = 0, 1, 2
K_ROLLOUT, K_GAE, K_SHUFFLE, K_ADV_STATS, K_UPDATE, K_REDUCE, K_OPTIM, K_ALLREDUCE, K_PACK, K_PERMUTE = range(10)
KERNEL_NAMES = ["rollout", "gae", "shuffle", "adv_stats", "update", "reduce", "optim", "allreduce", "pack", "permute"]

_lib = None


def load():
    """Loads the in-tree HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrlError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, fp, ip, dp, u8p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    L.crl_version.restype = C.c_int32
    L.crl_last_error.restype = C.c_char_p
    L.crl_device_count.argtypes = [ip]
    L.crl_ppo_create.argtypes = [C.POINTER(CrlConfig), C.c_int32, C.POINTER(vp)]
    L.crl_ppo_destroy.argtypes = [vp]
    L.crl_ppo_param_count.argtypes = [vp, C.POINTER(C.c_int64)]
    L.crl_sync.argtypes = [vp]
    L.crl_ppo_write.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.crl_ppo_read.argtypes = [vp, C.c_int32, vp, C.c_size_t]
    L.crl_policy_act.argtypes = [vp, fp, dp, C.c_int32, ip, fp, fp]
    L.crl_logprob_actions.argtypes = [vp, fp, ip, C.c_int32, fp, fp]
    L.crl_gae.argtypes = [C.c_int32, fp, fp, u8p, fp, u8p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, fp, fp]
    L.crl_gae_opt.argtypes = L.crl_gae.argtypes + [C.c_int32] * 3
    L.crl_rollout_store.argtypes = [vp, C.c_int32, fp, ip, fp, fp, u8p, fp]
    L.crl_env_reset.argtypes = [vp]
    L.crl_rollout_run.argtypes = [vp]
    L.crl_episode_stats_read.argtypes = [vp, C.POINTER(CrlEpisodeStats)]
    L.crl_compute_gae.argtypes = [vp]
    L.crl_shuffle.argtypes = [vp, C.c_uint64]
    L.crl_adv_stats.argtypes = [vp]
    L.crl_ppo_update_minibatch.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.POINTER(CrlStats)]
    L.crl_ppo_iterate.argtypes = [vp, C.c_int32, C.POINTER(CrlStats)]
    L.crl_ppo_iteration.argtypes = [vp, C.POINTER(C.c_int64)]
    L.crl_comm_unique_id.argtypes = [u8p]
    L.crl_comm_init.argtypes = [vp, u8p, C.c_int32, C.c_int32]
    L.crl_comm_init_external.argtypes = [vp, C.c_int32, C.c_int32]
    L.crl_comm_peer_export.argtypes = [vp, C.c_int32, C.c_int32, u8p]
    L.crl_comm_peer_attach.argtypes = [vp, u8p]
    L.crl_adv_stats_local.argtypes = [vp]
    L.crl_adv_stats_finish.argtypes = [vp]
    L.crl_prof_enable.argtypes = [vp, C.c_int32]
    L.crl_prof_read.argtypes = [vp, C.c_int32, dp, C.POINTER(C.c_int64)]
    L.crl_prof_reset.argtypes = [vp]
    i64p = C.POINTER(C.c_int64)
    L.crl_ppo_exact_reruns.argtypes = [vp, i64p]
    L.crl_episode_ring_enable.argtypes = [vp, C.c_int32]
    L.crl_episode_ring_read.argtypes = [vp, C.POINTER(CrlEpisodeRecord), C.c_int32, ip, i64p]
    L.crl_comm_destroy.argtypes = [vp]
    L.crl_ppo_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.crl_ppo_get_option.argtypes = [vp, C.c_char_p, i64p]
    L.crl_ppo_option_name.argtypes = [C.c_int32, C.POINTER(C.c_char_p), i64p]
    L.crl_ppo_option_count.argtypes = [ip]
    L.crl_gae_bench.argtypes = [C.c_int32] * 8 + [dp, dp]
    L.crl_a2c_create.argtypes = [C.POINTER(CrlA2CConfig), C.c_int32, C.POINTER(vp)]
    L.crl_a2c_destroy.argtypes = [vp]
    L.crl_a2c_param_count.argtypes = [vp, i64p]
    L.crl_a2c_write_params.argtypes = [vp, fp, C.c_size_t]
    L.crl_a2c_read_params.argtypes = [vp, fp, C.c_size_t]
    L.crl_a2c_read_env.argtypes = [vp, dp, i64p, ip]
    L.crl_a2c_read_buffer.argtypes = [vp, dp, ip, dp, u8p, C.c_int32]
    L.crl_a2c_run_until_update.argtypes = [vp, C.c_int64, C.POINTER(CrlA2CTrainStats), C.POINTER(CrlA2CEpisode), C.c_int32, ip, i64p]
    L.crl_a2c_discounted_future_rewards.argtypes = [C.c_int32, dp, u8p, C.c_int32, C.c_double, C.c_double, dp]
    L.crl_dqn_create.argtypes = [C.POINTER(CrlDQNConfig), C.c_int32, C.POINTER(vp)]
    L.crl_dqn_destroy.argtypes = [vp]
    L.crl_dqn_write_params.argtypes = [vp, fp, C.c_size_t]
    L.crl_dqn_read_params.argtypes = [vp, fp, fp, C.c_size_t]
    L.crl_dqn_status_read.argtypes = [vp, C.POINTER(CrlDQNStatus)]
    L.crl_dqn_run.argtypes = [vp, C.c_int64, C.POINTER(CrlDQNEpisode), C.c_int32, ip, C.POINTER(CrlDQNLossRecord), C.c_int32, ip, i64p]
    L.crl_dqn_q_values.argtypes = [vp, dp, C.c_int32, dp]
    L.crl_make_actor_critic.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, fp, C.c_size_t]
    L.crl_ppo_init_params.argtypes = [vp, C.c_uint64]
    L.crl_a2c_init_params.argtypes = [vp, C.c_uint64]
    L.crl_dqn_make_nn.argtypes = [C.c_uint64, fp, C.c_size_t]
    L.crl_dqn_init_params.argtypes = [vp, C.c_uint64]
    L.crl_comm_info.argtypes = [C.c_char_p, C.c_size_t, ip]
    L.crl_clock_probe.argtypes = [C.c_int32, C.c_double, dp, dp, dp]
    L.crl_ppo_iterate_async.argtypes = [vp, C.POINTER(CrlIterationReport), C.POINTER(CrlStats), C.POINTER(CrlEpisodeRecord), C.c_int32]
    L.crl_ppo_drain.argtypes = [vp, C.POINTER(CrlIterationReport), C.POINTER(CrlStats), C.POINTER(CrlEpisodeRecord), C.c_int32]
    L.crl_product_probe.argtypes = [C.c_int32, C.c_int32, fp, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, fp, fp]
    for name in EXPORTS:
        if name not in ("crl_version", "crl_last_error"):
            getattr(L, name).restype = C.c_int32
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise CrlError(load().crl_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int32(0)
    rc = load().crl_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def product_probe(flavour, A, B, chunks=1, scale_a=1.0, scale_b=1.0, col_scale=None, device=0):
    """crl_product_probe: A [rows, K] (C order), B [cols, K] (C order) float32 → C [chunks, cols, rows] float32 partial products."""
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
    rows, K = A.shape; cols = B.shape[0]
    out = np.zeros((chunks, cols, rows), np.float32)
    cs = None if col_scale is None else np.ascontiguousarray(col_scale, np.float32)
    check(load().crl_product_probe(device, flavour, _ptr(A, C.c_float), _ptr(B, C.c_float), rows, cols, K, chunks, scale_a, scale_b,
                                   None if cs is None else _ptr(cs, C.c_float), _ptr(out, C.c_float)))
    return out


def clock_probe(device=0, span_ms=5.0):
    """crl_clock_probe: shader clock under vector load, MHz (median, min, max over all waves of a chip-filling launch)."""
    med, lo, hi = C.c_double(0), C.c_double(0), C.c_double(0)
    check(load().crl_clock_probe(device, span_ms, C.byref(med), C.byref(lo), C.byref(hi)))
    return med.value, lo.value, hi.value


def comm_info():
    """crl_comm_info: (path of the librccl the library's all-reduce resolves to, ncclGetVersion code). Raises when librccl cannot be loaded."""
    buf = C.create_string_buffer(4096)
    ver = C.c_int32(0)
    check(load().crl_comm_info(buf, len(buf), C.byref(ver)))
    return buf.value.decode("utf-8", "replace"), int(ver.value)


def make_actor_critic_host(obs_dim, n_act, hidden, seed=0):
    """crl_make_actor_critic: the library's host-side restatement of Networks.make_actor_critic (networks.jl:36-49) — the start
    crl_ppo_init_params gives a handle; runs without a GPU."""
    n = 2 * (hidden * obs_dim + hidden + hidden * hidden + hidden) + n_act * hidden + n_act + hidden + 1
    out = np.zeros(n, np.float32)
    check(load().crl_make_actor_critic(obs_dim, n_act, hidden, seed, _ptr(out, C.c_float), n))
    return out


def dqn_make_nn_host(seed=0):
    """crl_dqn_make_nn: make_nn(env) of dqn.jl:22-26 (glorot-uniform weights, zero biases), flat in Flux.params order."""
    out = np.zeros(DQN_PARAM_COUNT, np.float32)
    check(load().crl_dqn_make_nn(seed, _ptr(out, C.c_float), out.size))
    return out


_FIELD_DTYPES = {
    F_OBS: np.float32, F_ACTION: np.int32, F_LOGPROB: np.float32, F_REWARD: np.float32, F_TERMINAL: np.uint8,
    F_VALUE: np.float32, F_ADVANTAGE: np.float32, F_RETURN: np.float32, F_PERM: np.int32, F_PARAMS: np.float32,
    F_GRADS: np.float32, F_ADAM_M: np.float32, F_ADAM_V: np.float32, F_ENV_STATE: np.float32, F_CUR_OBS: np.float32,
    F_NEXT_DONE: np.uint8, F_ENV_T: np.int32, F_BETAP: np.float64, F_ADV_SUMS: np.float64,
}


class Handle:
    """Owns one crl_ppo (one GPU / one data-parallel shard)."""

    def __init__(self, cfg: CrlConfig, device: int = 0):
        self.cfg = cfg
        self._h = C.c_void_p()
        check(load().crl_ppo_create(C.byref(cfg), device, C.byref(self._h)))
        n = C.c_int64()
        check(load().crl_ppo_param_count(self._h, C.byref(n)))
        self.P = n.value
        self.nt, self.k, self.d, self.A = cfg.num_envs, cfg.num_steps, cfg.obs_dim, cfg.n_act
        self.B = self.nt * self.k

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load().crl_ppo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, f):
        nt, k, d, B, P = self.nt, self.k, self.d, self.B, self.P
        return {F_OBS: (d, nt, k), F_PERM: (B,), F_PARAMS: (P,), F_GRADS: (P,), F_ADAM_M: (P,), F_ADAM_V: (P,),
                F_ENV_STATE: (d, nt), F_CUR_OBS: (d, nt), F_NEXT_DONE: (nt,), F_ENV_T: (nt,), F_BETAP: (24,),
                F_ADV_SUMS: (2 * self.cfg.num_minibatches,)}.get(f, (nt, k))

    def read(self, f):
        out = np.zeros(self._shape(f), _FIELD_DTYPES[f], order="F")
        check(load().crl_ppo_read(self._h, f, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def write(self, f, arr):
        a = np.asfortranarray(arr, _FIELD_DTYPES[f])
        if a.shape != self._shape(f):
            a = a.reshape(self._shape(f), order="F")
        a = np.asfortranarray(a)
        check(load().crl_ppo_write(self._h, f, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def sync(self):
        check(load().crl_sync(self._h))

    def init_params(self, seed=0):
        """ppo.jl:87 through the library's own initialiser (crl_ppo_init_params)."""
        check(load().crl_ppo_init_params(self._h, seed))

    def policy_act(self, obs, u, with_value=True):
        obs = np.asfortranarray(obs, np.float32)
        n = obs.shape[1] if obs.ndim == 2 else 1
        u = np.ascontiguousarray(u, np.float64)
        action = np.zeros(n, np.int32); logprob = np.zeros(n, np.float32); value = np.zeros(n, np.float32)
        check(load().crl_policy_act(self._h, _ptr(obs, C.c_float), _ptr(u, C.c_double), n, _ptr(action, C.c_int32),
                                    _ptr(logprob, C.c_float), _ptr(value, C.c_float) if with_value else None))
        return action, logprob, value

    def logprob_actions(self, obs, actions):
        obs = np.asfortranarray(obs, np.float32)
        n = obs.shape[1]
        actions = np.ascontiguousarray(actions, np.int32)
        logprob = np.zeros(n, np.float32); ent = np.zeros((self.A, n), np.float32, order="F")
        check(load().crl_logprob_actions(self._h, _ptr(obs, C.c_float), _ptr(actions, C.c_int32), n, _ptr(logprob, C.c_float),
                                         _ptr(ent, C.c_float)))
        return logprob, ent

    def rollout_store(self, step, obs, action, logprob, reward, terminal, value):
        obs = np.asfortranarray(obs, np.float32); action = np.ascontiguousarray(action, np.int32)
        logprob = np.ascontiguousarray(logprob, np.float32); reward = np.ascontiguousarray(reward, np.float32)
        terminal = np.ascontiguousarray(terminal, np.uint8); value = np.ascontiguousarray(value, np.float32)
        check(load().crl_rollout_store(self._h, step, _ptr(obs, C.c_float), _ptr(action, C.c_int32), _ptr(logprob, C.c_float),
                                       _ptr(reward, C.c_float), _ptr(terminal, C.c_uint8), _ptr(value, C.c_float)))

    def env_reset(self):
        check(load().crl_env_reset(self._h))

    def rollout_run(self):
        check(load().crl_rollout_run(self._h))

    def episode_stats(self):
        st = CrlEpisodeStats()
        check(load().crl_episode_stats_read(self._h, C.byref(st)))
        return {"episodes": st.episodes, "return_sum": st.return_sum, "length_sum": st.length_sum, "return_max": st.return_max}

    def compute_gae(self):
        check(load().crl_compute_gae(self._h))

    def shuffle(self, epoch_id):
        check(load().crl_shuffle(self._h, epoch_id))

    def adv_stats(self):
        check(load().crl_adv_stats(self._h))

    def update_minibatch(self, mb, eta, apply_update=True, want_stats=True):
        st = CrlStats()
        check(load().crl_ppo_update_minibatch(self._h, mb, eta, int(apply_update), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def iterate(self, n_iters=1, want_stats=True):
        n = self.cfg.update_epochs * self.cfg.num_minibatches
        arr = (CrlStats * n)()
        check(load().crl_ppo_iterate(self._h, n_iters, arr if want_stats else None))
        return [a.as_dict() for a in arr] if want_stats else None

    def _report(self, fn, want_stats):
        n = self.cfg.update_epochs * self.cfg.num_minibatches
        cap = getattr(self, "_ring_cap", 0)
        rep = CrlIterationReport(); arr = (CrlStats * n)(); ring = (CrlEpisodeRecord * max(cap, 1))()
        check(fn(self._h, C.byref(rep), arr if want_stats else None, ring if cap else None, cap))
        if rep.iteration < 0:
            return None
        recs = sorted((ring[i].step, ring[i].env, ring[i].episode_return, ring[i].episode_length) for i in range(rep.n_ring))
        return {"iteration": rep.iteration, "stats": [a.as_dict() for a in arr] if want_stats else None,
                "episodes": {"episodes": rep.episodes.episodes, "return_sum": rep.episodes.return_sum, "length_sum": rep.episodes.length_sum,
                             "return_max": rep.episodes.return_max},
                "records": recs, "n_episodes": rep.n_episodes}

    def iterate_async(self, want_stats=True):
        """crl_ppo_iterate_async: enqueues one iteration and returns the report of the iteration BEFORE it (None on the first call): its loss records,
        episode statistics and per-episode records (sorted by (step, env) like episode_records()) — read without making the GPU wait for the host."""
        return self._report(load().crl_ppo_iterate_async, want_stats)

    def drain(self, want_stats=True):
        """crl_ppo_drain: the report of the last iteration enqueued through iterate_async (None: nothing pending)."""
        return self._report(load().crl_ppo_drain, want_stats)

    @property
    def iteration(self):
        it = C.c_int64()
        check(load().crl_ppo_iteration(self._h, C.byref(it)))
        return it.value

    def episode_ring_enable(self, capacity):
        check(load().crl_episode_ring_enable(self._h, int(capacity)))
        self._ring_cap = int(capacity)

    def episode_records(self):
        """(records sorted by (step, env) — the reference's logging order —, number of episodes that ended)"""
        cap = getattr(self, "_ring_cap", 0)
        buf = (CrlEpisodeRecord * max(cap, 1))(); n = C.c_int32(); tot = C.c_int64()
        check(load().crl_episode_ring_read(self._h, buf, cap, C.byref(n), C.byref(tot)))
        recs = [(buf[i].step, buf[i].env, buf[i].episode_return, buf[i].episode_length) for i in range(n.value)]
        recs.sort()
        return recs, tot.value

    @property
    def exact_reruns(self):
        n = C.c_int64()
        check(load().crl_ppo_exact_reruns(self._h, C.byref(n)))
        return n.value

    def comm_init(self, unique_id: bytes, world_size: int, rank: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(load().crl_comm_init(self._h, buf, world_size, rank))

    def comm_peer_export(self, world_size: int, rank: int) -> bytes:
        """Allocates this rank's mailbox for the one-shot peer all-reduce; returns its 64-byte IPC handle."""
        buf = (C.c_uint8 * 64)()
        check(load().crl_comm_peer_export(self._h, world_size, rank, buf))
        return bytes(buf)

    def comm_peer_attach(self, handles: bytes):
        """handles: world_size x 64 bytes in rank order (every rank's crl_comm_peer_export result)."""
        buf = (C.c_uint8 * len(handles)).from_buffer_copy(handles)
        check(load().crl_comm_peer_attach(self._h, buf))

    def comm_init_external(self, world_size: int, rank: int):
        check(load().crl_comm_init_external(self._h, world_size, rank))

    def adv_stats_local(self):
        check(load().crl_adv_stats_local(self._h))

    def adv_stats_finish(self):
        check(load().crl_adv_stats_finish(self._h))

    def comm_destroy(self):
        check(load().crl_comm_destroy(self._h))

    def set_option(self, key: str, value: int):
        """crl_ppo_set_option: per-handle kernel-flavour / numerics switches (include/cleanrl_hip.h lists them)."""
        check(load().crl_ppo_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        check(load().crl_ppo_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def options(self) -> dict:
        return {k: self.get_option(k) for k in option_names()}

    def prof_enable(self, on=True):
        check(load().crl_prof_enable(self._h, int(on)))

    def prof_reset(self):
        check(load().crl_prof_reset(self._h))

    def prof_read(self):
        out = {}
        for kid, name in enumerate(KERNEL_NAMES):
            ms = C.c_double(); n = C.c_int64()
            check(load().crl_prof_read(self._h, kid, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out


def option_names():
    """Names of every option crl_ppo_set_option accepts, in table order."""
    n = C.c_int32()
    check(load().crl_ppo_option_count(C.byref(n)))
    out = []
    for i in range(n.value):
        name = C.c_char_p(); d = C.c_int64()
        check(load().crl_ppo_option_name(i, C.byref(name), C.byref(d)))
        out.append(name.value.decode())
    return out


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * 128)()
    check(load().crl_comm_unique_id(buf))
    return bytes(buf)


def gae_host(value, reward, terminal, next_value, next_done, gamma, lam, mode=GAE_COMPAT, device=0, seg=0, tile=0, nt_loads=2):
    """crl_gae / crl_gae_opt on host arrays: value/reward/terminal are (nt,k) Fortran-ordered; seg / tile / nt_loads select the kernel
    flavour like the handle options gae_seg / gae_tile / gae_nt_loads (defaults: the library decides)."""
    value = np.asfortranarray(value, np.float32); reward = np.asfortranarray(reward, np.float32)
    terminal = np.asfortranarray(terminal, np.uint8)
    nt, k = value.shape
    nv = None if next_value is None else np.ascontiguousarray(next_value, np.float32)
    nd = None if next_done is None else np.ascontiguousarray(next_done, np.uint8)
    adv = np.zeros((nt, k), np.float32, order="F"); ret = np.zeros((nt, k), np.float32, order="F")
    args = (device, _ptr(value, C.c_float), _ptr(reward, C.c_float), _ptr(terminal, C.c_uint8),
            _ptr(nv, C.c_float) if nv is not None else None, _ptr(nd, C.c_uint8) if nd is not None else None,
            nt, k, gamma, lam, mode, _ptr(adv, C.c_float), _ptr(ret, C.c_float))
    if (seg, tile, nt_loads) == (0, 0, 2):
        check(load().crl_gae(*args))
    else:
        check(load().crl_gae_opt(*args, seg, tile, nt_loads))
    return adv, ret


def gae_bench(nt, k=128, seg=0, tile=0, nt_loads=0, flush_mb=0, reps=10, device=0):
    """crl_gae_bench: (gae launch times, same-byte-count float4 copy launch times), ms each, `reps` of them."""
    g = np.zeros(reps, np.float64); c = np.zeros(reps, np.float64)
    check(load().crl_gae_bench(device, nt, k, seg, tile, nt_loads, flush_mb, reps, _ptr(g, C.c_double), _ptr(c, C.c_double)))
    return g, c


class A2CHandle:
    """Owns one crl_a2c* (include/cleanrl_hip.h, A2C block): networks, optimiser state, replay buffer and env on one GPU."""

    def __init__(self, cfg: CrlA2CConfig, device=0):
        self._L = load()
        self.cfg = cfg
        self._h = C.c_void_p()
        check(self._L.crl_a2c_create(C.byref(cfg), device, C.byref(self._h)))
        n = C.c_int64()
        check(self._L.crl_a2c_param_count(self._h, C.byref(n)))
        self.param_count = n.value

    def close(self):
        if self._h:
            self._L.crl_a2c_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        check(self._L.crl_a2c_write_params(self._h, _ptr(p, C.c_float), p.size))

    def init_params(self, seed=0):
        check(self._L.crl_a2c_init_params(self._h, seed))

    def read_params(self):
        p = np.zeros(self.param_count, np.float32)
        check(self._L.crl_a2c_read_params(self._h, _ptr(p, C.c_float), p.size))
        return p

    def env(self):
        s = np.zeros(4, np.float64); g = C.c_int64(); n = C.c_int32()
        check(self._L.crl_a2c_read_env(self._h, _ptr(s, C.c_double), C.byref(g), C.byref(n)))
        return s, g.value, n.value

    def buffer(self):
        cap = 2 * self.cfg.min_replay_size
        st = np.zeros((4, cap), np.float64, order="F"); a = np.zeros(cap, np.int32); r = np.zeros(cap, np.float64); t = np.zeros(cap, np.uint8)
        check(self._L.crl_a2c_read_buffer(self._h, _ptr(st, C.c_double), _ptr(a, C.c_int32), _ptr(r, C.c_double), _ptr(t, C.c_uint8), cap))
        n = self.env()[2]
        return st[:, :n], a[:n], r[:n], t[:n]

    def run_until_update(self, max_env_steps=1 << 40, max_eps=4096):
        ts = CrlA2CTrainStats(); eps = (CrlA2CEpisode * max_eps)(); n = C.c_int32(); taken = C.c_int64()
        check(self._L.crl_a2c_run_until_update(self._h, max_env_steps, C.byref(ts), eps, max_eps, C.byref(n), C.byref(taken)))
        episodes = [(eps[i].episode_return, eps[i].episode_length, eps[i].global_step) for i in range(n.value)]
        return taken.value, dict(actor_loss=ts.actor_loss, critic_loss=ts.critic_loss, n=ts.n, trained=bool(ts.trained)), episodes


def a2c_discounted_future_rewards_host(rewards, terminals, final_value, gamma, device=0):
    r = np.ascontiguousarray(rewards, np.float64); t = np.ascontiguousarray(terminals, np.uint8)
    out = np.zeros(r.size, np.float64)
    check(load().crl_a2c_discounted_future_rewards(device, _ptr(r, C.c_double), _ptr(t, C.c_uint8), r.size, float(final_value),
                                                   float(gamma), _ptr(out, C.c_double)))
    return out


class DQNHandle:
    """Owns one crl_dqn* (include/cleanrl_hip.h, DQN block)."""

    def __init__(self, cfg: CrlDQNConfig, device=0):
        self._L = load()
        self.cfg = cfg
        self._h = C.c_void_p()
        check(self._L.crl_dqn_create(C.byref(cfg), device, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.crl_dqn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        check(self._L.crl_dqn_write_params(self._h, _ptr(p, C.c_float), p.size))

    def init_params(self, seed=0):
        check(self._L.crl_dqn_init_params(self._h, seed))

    def read_params(self):
        q = np.zeros(DQN_PARAM_COUNT, np.float32); t = np.zeros(DQN_PARAM_COUNT, np.float32)
        check(self._L.crl_dqn_read_params(self._h, _ptr(q, C.c_float), _ptr(t, C.c_float), q.size))
        return q, t

    def status(self):
        st = CrlDQNStatus()
        check(self._L.crl_dqn_status_read(self._h, C.byref(st)))
        return dict(state=np.array(list(st.env_state)), global_step=st.global_step, rb_size=st.rb_size, n_updates=st.n_updates,
                    last_loss=st.last_loss)

    def run(self, max_env_steps, max_eps=8192, max_losses=4096):
        eps = (CrlDQNEpisode * max_eps)(); ls = (CrlDQNLossRecord * max_losses)(); ne = C.c_int32(); nl = C.c_int32(); taken = C.c_int64()
        check(self._L.crl_dqn_run(self._h, max_env_steps, eps, max_eps, C.byref(ne), ls, max_losses, C.byref(nl), C.byref(taken)))
        return (taken.value, [(eps[i].episode_return, eps[i].episode_length, eps[i].global_step, eps[i].epsilon) for i in range(ne.value)],
                [(ls[i].global_step, ls[i].loss) for i in range(nl.value)])

    def q_values(self, obs):
        obs = np.asfortranarray(obs, np.float64)
        if obs.ndim == 1:
            obs = obs[:, None]
        n = obs.shape[1]
        q = np.zeros((2, n), np.float64, order="F")
        check(self._L.crl_dqn_q_values(self._h, _ptr(obs, C.c_double), n, _ptr(q, C.c_double)))
        return q
