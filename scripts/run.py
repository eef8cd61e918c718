#!/usr/bin/env python3
"""Runner in the style the reference README plans (`experiments/run_ppo`): every field of the algorithm's config struct is a
command line option (ConfigParser.argparse_struct), records go to the "CleanRL" logger sinks (Logger.make_logger).

    python scripts/run.py ppo --num_envs 4096 --num_steps 128 --total_timesteps 10485760
    python scripts/run.py a2c --total_timesteps 100000
    python scripts/run.py dqn --total_timesteps 50000
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl  # noqa: E402
from cleanrl_jl_amd import config_parser  # noqa: E402


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in ("ppo", "a2c", "dqn"):
        raise SystemExit(__doc__)
    algo, argv = sys.argv[1], sys.argv[2:]
    if algo == "ppo":
        crl.ppo(config_parser.argparse_struct(crl.PPOConfig(), argv), logger_kw=dict(to_terminal=True, to_tensorboard=False))
    elif algo == "a2c":
        crl.a2c(config_parser.argparse_struct(crl.A2CConfig(), argv), to_terminal=True, to_tensorboard=False).close()
    else:
        crl.dqn(config_parser.argparse_struct(crl.DQNConfig(), argv), to_terminal=True, to_tensorboard=False).close()


if __name__ == "__main__":
    main()
