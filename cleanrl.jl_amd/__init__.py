"""cleanrl.jl_amd — MI355X-native PPO rollout + GAE + update hot path behind the API of sash-a/CleanRL.jl's
src/algorithms/ppo.jl. Import as `cleanrl_jl_amd` (root shim) — the directory name is not a Python identifier.

Layout: csrc/ (HIP kernels + the C ABI of include/cleanrl_hip.h) and the host-side mirror of the reference interface.
"""
from . import _lib, config_parser
from ._lib import CrlError, Handle, comm_unique_id, device_count
from .logger import make_logger
from .networks import make_actor_critic
from .a2c import A2CAgent, A2CConfig, a2c, discounted_future_rewards
from .dqn import DQNAgent, DQNConfig, dqn, linear_schedule, make_nn
from .ppo import Agent, Policy, PPOConfig, gae, get_action, logprob_actions, ppo, train

__all__ = ["_lib", "config_parser", "CrlError", "Handle", "comm_unique_id", "device_count", "make_logger", "make_actor_critic", "Agent",
           "Policy", "PPOConfig", "gae", "get_action", "logprob_actions", "ppo", "train", "A2CAgent", "A2CConfig", "a2c",
           "discounted_future_rewards", "DQNAgent", "DQNConfig", "dqn", "linear_schedule", "make_nn"]
