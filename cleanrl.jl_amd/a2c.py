"""Host-side mirror of the reference's A2C file (src/algorithms/a2c.jl) over the C ABI: same names (`A2CConfig`, `a2c`,
`discounted_future_rewards`), same argument meaning and the same two logger records. All arithmetic runs in
libcleanrl_hip.so (csrc/a2c.hip); this module only orchestrates and logs."""
import dataclasses
import time

import numpy as np

from . import _lib as L
from . import logger as Logger
from . import networks


@dataclasses.dataclass
class A2CConfig:
    """a2c.jl:1-10 (field names and defaults of the reference)."""
    run_name: str = dataclasses.field(default_factory=lambda: time.strftime("%y-%m-%d|%H:%M:%S"))
    lr: float = 0.0001
    total_timesteps: int = 1_000_000
    min_replay_size: int = 512
    gamma: float = 0.99


def discounted_future_rewards(rewards, terminals, final_value, gamma, *, device=0):
    """a2c.jl:13-24. All-Float64 like the reference's method signature; `terminals` is a Bool vector."""
    rewards = np.asarray(rewards)
    if rewards.dtype != np.float64:
        raise TypeError("discounted_future_rewards: rewards must be Float64 (a2c.jl:13 Vector{T}, final_value::T, γ::T)")
    terminals = np.asarray(terminals)
    if terminals.shape != rewards.shape:
        raise ValueError("discounted_future_rewards: rewards and terminals differ in length")
    return L.a2c_discounted_future_rewards_host(rewards, terminals.astype(np.uint8), float(final_value), float(gamma), device)


class A2CAgent:
    """Actor, critic, Optimiser(ClipNorm(0.5), Adam(lr)) state, ReplayBuffer(2*min_replay_size) and the CartPoleEnv of one
    a2c(config) run (a2c.jl:32-52), resident on one GPU."""

    def __init__(self, config: A2CConfig, *, device=0, params=None, seed=0x5EED, init_seed=0, max_steps=500):
        self.config = config
        self.crl_cfg = L.CrlA2CConfig(float(config.lr), int(config.total_timesteps), int(config.min_replay_size), int(max_steps),
                                      float(config.gamma), seed)
        self.handle = L.A2CHandle(self.crl_cfg, device)
        if params is None:
            params = networks.make_actor_critic(2, 4, [64, 64], seed=init_seed)   # a2c.jl:37 make_actor_critic(env)
        self.handle.write_params(params)

    def close(self):
        self.handle.close()


def a2c(config: A2CConfig = None, *, device=0, seed=0x5EED, params=None, **logger_kw):
    """a2c.jl:29-113. Returns the agent after total_timesteps env steps. The "CleanRL" logger receives the reference's
    "Training Statistics" (actor_loss, critic_loss; a2c.jl:100) and "Episode Statistics" (a2c.jl:106) records."""
    import logging
    config = config or A2CConfig()
    Logger.make_logger(f"a2c|{config.run_name}", **({"to_terminal": False} | logger_kw))      # a2c.jl:30
    lg = logging.getLogger("CleanRL")

    def log(msg, **kv):
        lg.info(msg, extra={"crl": kv})
    agent = A2CAgent(config, device=device, seed=seed, params=params)
    start = time.time()
    while True:
        taken, ts, episodes = agent.handle.run_until_update()
        # the episode whose end triggered the update is the call's LAST record, and the reference logs the update first (a2c.jl:100, then :106)
        for i, (ret, length, gstep) in enumerate(episodes):                   # a2c.jl:105-106
            if ts["trained"] and i == len(episodes) - 1:
                log("Training Statistics", actor_loss=ts["actor_loss"], critic_loss=ts["critic_loss"])   # a2c.jl:100
            log("Episode Statistics", episode_return=ret, episode_length=length, global_step=gstep,
                steps_per_sec=int(gstep / max(time.time() - start, 1e-9)))
        if ts["trained"] and not episodes:
            log("Training Statistics", actor_loss=ts["actor_loss"], critic_loss=ts["critic_loss"])
        if taken == 0 or agent.handle.env()[1] >= config.total_timesteps:
            break
    return agent
