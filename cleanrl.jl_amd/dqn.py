"""Host-side mirror of the reference's DQN file (src/algorithms/dqn.jl) over the C ABI: same names (`DQNConfig`, `make_nn`,
`linear_schedule`, `dqn`) and logger records. All arithmetic of the loop runs in libcleanrl_hip.so (csrc/dqn.hip)."""
import dataclasses
import logging
import time

import numpy as np

from . import _lib as L
from . import logger as Logger


@dataclasses.dataclass
class DQNConfig:
    """dqn.jl:1-19 (field names — including `log_frequencey` — and defaults of the reference)."""
    run_name: str = dataclasses.field(default_factory=lambda: time.strftime("%y-%m-%d|%H:%M:%S"))
    log_frequencey: int = 1000
    total_timesteps: int = 500_000
    buffer_size: int = 10_000
    min_buff_size: int = 200
    lr: float = 0.0001
    train_freq: int = 10
    target_net_freq: int = 100
    batch_size: int = 120
    gamma: float = 0.99
    epsilon_start: float = 1.0
    epsilon_end: float = 0.05
    epsilon_duration: float = 10_000


def linear_schedule(start_e, end_e, duration, t):
    """dqn.jl:28-31 (host-side helper; the loop evaluates the same expression on the device)."""
    slope = (end_e - start_e) / duration
    return max(slope * t + start_e, end_e)


def make_nn(seed=0):
    """dqn.jl:22-26 for CartPole: Chain(Dense(4,120,relu), Dense(120,84,relu), Dense(84,2)) as ONE flat float32 vector in
    Flux.params order; Flux's default init (glorot_uniform weights, zero biases) with a numpy stream — weights are an input."""
    rng = np.random.default_rng(seed)
    out = np.zeros(L.DQN_PARAM_COUNT, np.float32)
    off = np.cumsum([0, 480, 120, 10080, 84, 168, 2])
    for i, (rows, cols) in zip((0, 2, 4), ((120, 4), (84, 120), (2, 84))):
        lim = np.sqrt(6.0 / (rows + cols))
        out[off[i]:off[i + 1]] = rng.uniform(-lim, lim, rows * cols).astype(np.float32)
    return out


class DQNAgent:
    """q_net, target_net, Adam state, ReplayBuffer(buffer_size) and the CartPoleEnv of one dqn(config) run, on one GPU."""

    def __init__(self, config: DQNConfig, *, device=0, params=None, seed=0x5EED, init_seed=0, max_steps=200):
        self.config = config
        self.crl_cfg = L.CrlDQNConfig(int(config.log_frequencey), int(config.total_timesteps), int(config.buffer_size),
                                      int(config.min_buff_size), float(config.lr), int(config.train_freq), int(config.target_net_freq),
                                      int(config.batch_size), float(config.gamma), float(config.epsilon_start), float(config.epsilon_end),
                                      float(config.epsilon_duration), int(max_steps), 0, seed)
        self.handle = L.DQNHandle(self.crl_cfg, device)
        self.handle.write_params(make_nn(init_seed) if params is None else params)

    def close(self):
        self.handle.close()


def dqn(config: DQNConfig = None, *, device=0, seed=0x5EED, params=None, chunk=10_000, **logger_kw):
    """dqn.jl:34-120. One library call per `chunk` env steps; the "CleanRL" logger receives "Episode Statistics"
    (episode_return, episode_length, global_step, ϵ, steps_per_sec; dqn.jl:88) and "Training Statistics" (loss; dqn.jl:116)."""
    config = config or DQNConfig()
    Logger.make_logger(f"dqn|{config.run_name}", **({"to_terminal": False} | logger_kw))         # dqn.jl:35
    lg = logging.getLogger("CleanRL")
    agent = DQNAgent(config, device=device, seed=seed, params=params)
    start = time.time()
    while True:
        taken, episodes, losses = agent.handle.run(chunk)
        records = [(g, "Episode Statistics", dict(episode_return=r, episode_length=n, global_step=g, **{"ϵ": e},
                                                  steps_per_sec=int(g / max(time.time() - start, 1e-9)))) for r, n, g, e in episodes]
        records += [(g, "Training Statistics", dict(loss=v)) for g, v in losses]
        for _, msg, kv in sorted(records, key=lambda x: x[0]):
            lg.info(msg, extra={"crl": kv})
        if taken == 0 or agent.handle.status()["global_step"] >= config.total_timesteps:
            break
    return agent
