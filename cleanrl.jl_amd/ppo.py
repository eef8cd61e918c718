"""Host-side mirror of sash-a/CleanRL.jl `src/algorithms/ppo.jl` over the C ABI (include/cleanrl_hip.h).

The reference host language is Julia, which this image lacks; the Julia shell a maintainer would use is in
INTEGRATION.md / julia/CleanRLHip.jl. This Python mirror keeps the same names, argument meaning and error behaviour
(`PPOConfig`, `ppo`, `get_action`, `logprob_actions`, `gae`, the two logger records) so the parity tests read like
tests of the reference. All arithmetic happens in libcleanrl_hip.so on the GPU; nothing here computes on the CPU.
"""
import dataclasses
import logging
import time

import numpy as np

from . import _lib as L
from . import networks

log = logging.getLogger("CleanRL")


@dataclasses.dataclass
class PPOConfig:
    """ppo.jl:1-19 — same field names and defaults (Float32 fields are rounded to float32 at the boundary)."""
    total_timesteps: int = 500_000
    num_steps: int = 32
    num_envs: int = 4
    num_minibatches: int = 4
    update_epochs: int = 4
    lr: float = 2.5e-4
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_coef: float = 0.2
    ent_coeff: float = 0.01
    v_coef: float = 0.5
    normalize_advantages: bool = True
    clip_value_loss: bool = True
    anneal_lr: bool = True


def _crl_config(config: PPOConfig, *, obs_dim=4, n_act=2, hidden=64, gae_mode=L.GAE_COMPAT, env_kind=L.ENV_CARTPOLE,
                stale_obs=True, env_id_offset=0, shuffle_mode=L.SHUFFLE_BLOCKED_FY, seed=0x5EED, num_envs=None):
    return L.CrlConfig(config.total_timesteps, config.num_steps, config.num_envs if num_envs is None else num_envs,
                       config.num_minibatches, config.update_epochs, config.lr, config.gamma, config.gae_lambda,
                       config.clip_coef, config.ent_coeff, config.v_coef, int(config.normalize_advantages),
                       int(config.clip_value_loss), int(config.anneal_lr), obs_dim, n_act, hidden, gae_mode, env_kind,
                       int(stale_obs), env_id_offset, shuffle_mode, seed)


class Policy:
    """What `actor` / `critic` (Flux Chains in the reference, networks.jl:36-49) are here: a view of one network of an
    Agent whose weights live in HBM."""

    def __init__(self, agent, which):
        self.agent, self.which = agent, which

    def __call__(self, obs):
        obs = np.asfortranarray(obs, np.float32)
        if obs.ndim == 1:
            obs = obs[:, None]
        if self.which == "critic":
            _, _, v = self.agent.handle.policy_act(obs, np.zeros(obs.shape[1]))
            return v[None, :]
        raise TypeError("call get_action / logprob_actions for the actor (logits stay on the GPU)")


class Agent:
    """Actor + critic + optimiser state + rollout buffer + vectorised env of one PPO run, all resident on one GPU."""

    def __init__(self, config: PPOConfig, *, device=0, params=None, seed=0x5EED, init_seed=0, options=None, **shape):
        self.config = config
        self.crl_cfg = _crl_config(config, seed=seed, **shape)
        self.handle = L.Handle(self.crl_cfg, device)
        for key, value in (options or {}).items():      # crl_ppo_set_option: kernel-flavour switches of this handle
            self.handle.set_option(key, value)
        if params is None:
            params = networks.make_actor_critic(self.crl_cfg.n_act, self.crl_cfg.obs_dim, [self.crl_cfg.hidden] * 2, seed=init_seed)
        self.set_params(params)
        self.actor, self.critic = Policy(self, "actor"), Policy(self, "critic")

    def set_params(self, flat):
        self.handle.write(L.F_PARAMS, np.ascontiguousarray(flat, np.float32))

    def get_params(self):
        return self.handle.read(L.F_PARAMS)

    def close(self):
        self.handle.close()


def get_action(obs, actor: Policy, u=None, rng=None):
    """ppo.jl:21-32. Returns (action, logprob_action); actions are 1-based like the reference's `Base.OneTo(2)`.
    `u` are the uniform Float64 draws StatsBase.sample would take from the global RNG (one per column)."""
    obs = np.asfortranarray(obs, np.float32)
    if obs.ndim == 1:
        obs = obs[:, None]
    n = obs.shape[1]
    if u is None:
        u = (rng or np.random.default_rng()).random(n)
    a, lp, _ = actor.agent.handle.policy_act(obs, u, with_value=False)
    return a.astype(np.int64) + 1, lp


def logprob_actions(obs, actor: Policy, actions):
    """ppo.jl:34-45. `actions` 1-based Int32 like the reference; entropy is the (n_act, batch) matrix (Q3)."""
    actions = np.asarray(actions)
    if actions.dtype != np.int32:
        raise TypeError("logprob_actions: actions must be Int32 (ppo.jl:34 AbstractVector{Int32})")
    return actor.agent.handle.logprob_actions(obs, actions - 1)


def gae(values, rewards, terminals, gamma, lam, *, mode=L.GAE_COMPAT, device=0, seg=0, tile=0, nt_loads=2):
    """ppo.jl:48-73 for one env: values [0,k], rewards [1,k], terminals [0,k] → advantages.
    In compat mode the last slot is 0.0 (the reference leaves it uninitialised, ppo.jl:62,66)."""
    values = np.asarray(values, np.float32); rewards = np.asarray(rewards, np.float32)
    terminals = np.asarray(terminals).astype(np.uint8)
    k = rewards.shape[0]
    if values.shape[0] != k + 1 or terminals.shape[0] != k + 1:
        raise ValueError("gae: values and terminals need length(rewards)+1 entries")
    if k == 0:
        return np.zeros(0, np.float32)
    adv, _ = L.gae_host(values[None, :k], rewards[None, :], terminals[None, :k], values[k:], terminals[k:], gamma, lam, mode, device, seg, tile, nt_loads)
    return adv[0]


def _linear_eta(config, update, num_updates):
    # ppo.jl:118-121 (update is 1-based); Float64 like the reference
    if not config.anneal_lr:
        return float(np.float32(config.lr))
    frac = 1.0 - (update - 1.0) / num_updates
    return frac * float(np.float32(config.lr))


def train(agent: Agent, num_updates=None, log_every=1, episode_records=0):
    """`train!`-style driver = the `for update in 1:num_updates` loop of ppo.jl:117-253, fully on device.
    Emits the reference's two records: "Episode Statistics" and "Training Statistics". By default the episode record is one
    aggregate per rollout (with 65536 envs the reference's one-record-per-episode is ~10^5 log lines per update);
    `episode_records=N` turns on the device ring (crl_episode_ring_enable) and logs up to N episodes per rollout one by one,
    in the reference's order (step, then env; global_step as in ppo.jl:124,148)."""
    cfg = agent.config
    batch_size = cfg.num_steps * cfg.num_envs
    if num_updates is None:
        num_updates = max(1, cfg.total_timesteps // batch_size)  # ppo.jl:91
    h = agent.handle
    if episode_records:
        h.episode_ring_enable(int(episode_records))
    if h.iteration == 0:
        h.env_reset()
    start_time = time.time()
    last_log_step = 0

    def emit(rep):
        """The records of one update, in the reference's order: its episodes (ppo.jl:147-165), then its 16 minibatches (ppo.jl:246-248)."""
        nonlocal last_log_step
        base = rep["iteration"] * batch_size
        global_step = base + batch_size
        ep = rep["episodes"]
        if episode_records:
            for step, env, ret, length in rep["records"]:
                gs = base + (step + 1) * cfg.num_envs                              # ppo.jl:124 global_step += num_envs
                inc = 0 if last_log_step == 0 else gs - last_log_step
                log.info("Episode Statistics", extra={"crl": dict(
                    episode_return=ret, episode_length=length, global_step=gs,
                    steps_per_sec=int(gs / max(time.time() - start_time, 1e-9)), log_step_increment=inc)})
                last_log_step = gs
        elif ep["episodes"] > 0:
            inc = 0 if last_log_step == 0 else global_step - last_log_step
            log.info("Episode Statistics", extra={"crl": dict(
                episode_return=ep["return_sum"] / ep["episodes"], episode_length=ep["length_sum"] / ep["episodes"],
                global_step=global_step, steps_per_sec=int(global_step / max(time.time() - start_time, 1e-9)), log_step_increment=inc)})
            last_log_step = global_step
        if log_every:
            for s in rep["stats"]:
                inc = 0 if last_log_step == 0 else global_step - last_log_step
                log.info("Training Statistics", extra={"crl": dict(
                    loss=s["loss"], pg_loss=s["pg_loss"], v_loss=s["v_loss"], entropy_loss=s["entropy_loss"],
                    log_step_increment=inc)})
                last_log_step = global_step

    # Pipelined read-back (crl_ppo_iterate_async): update k's records are picked up after update k + 1 has been enqueued, so the GPU never idles while the host
    # logs; the record stream is the same, one update late, and crl_ppo_drain hands over the last one.
    for _ in range(num_updates):
        rep = h.iterate_async(want_stats=bool(log_every))
        if rep is not None:
            emit(rep)
    rep = h.drain(want_stats=bool(log_every))
    if rep is not None:
        emit(rep)
    return agent


def ppo(config: PPOConfig = None, *, device=0, seed=0x5EED, init_seed=0, params=None, episode_records=4096, run_name="ppo-2-test",
        logger_kw=None, **shape):
    """ppo.jl:75 — `ppo(config::PPOConfig=PPOConfig())`: CartPole, 2x64 actor/critic, whole loop on one MI355X. Like the Julia shell
    (julia/CleanRLHip.jl) it logs ONE "Episode Statistics" record per finished episode in the reference's order (ppo.jl:147-165), up
    to `episode_records` per rollout (the device ring's capacity; 0 = one aggregate record per update), and the 16 "Training
    Statistics" records of every update (ppo.jl:246-248). `logger_kw` goes to Logger.make_logger (logger.jl:7); `shape` keywords
    (obs_dim, n_act, hidden, env_kind, gae_mode, stale_obs, shuffle_mode) to the Agent — the reference derives them from the env
    (ppo.jl:85-87)."""
    from . import logger as _logger
    config = config or PPOConfig()
    _logger.make_logger(run_name, **({"to_terminal": False} | (logger_kw or {})))
    agent = Agent(config, device=device, seed=seed, init_seed=init_seed, params=params, **shape)
    try:
        train(agent, episode_records=episode_records)
        return agent.get_params()
    finally:
        agent.close()
