# run_ppo.jl — the per-algorithm runner README.md:24 of the reference lists as a TODO ("Make individual file runners e.g
# experiments/run_ppo.(jl/sh)"): every field of PPOConfig (ppo.jl:1-19) becomes a `--field value` flag through the reference's own
# ConfigParser.argparse_struct (config_parser.jl:18-40) and the parsed struct goes to the MI355X loop.
#
#   julia --project=/path/to/CleanRL.jl julia/run_ppo.jl --num_envs 65536 --num_steps 128 --total_timesteps 83886080
#
# NOT EXECUTED IN THIS REPO'S CI (no Julia in the image, SURVEY.md F4); tests/test_host_cpu.py checks the names it uses against
# CleanRLHip.jl. The Python twin that does run here is scripts/run.py.
using CleanRL                                   # the reference package: ConfigParser (ArgParse) and Logger (TensorBoard / JSON / console sinks)
include(joinpath(@__DIR__, "CleanRLHip.jl"))
using .CleanRLHip

config = CleanRL.ConfigParser.argparse_struct(CleanRLHip.PPOConfig())
# ppo.jl:75-87: same entry point, same logger call, and the reference's OWN networks — Networks.make_actor_critic(...) .|> Flux.f32 (ppo.jl:87),
# flattened in Flux.params order. (CleanRLHip is included at top level here, so its in-package default cannot see CleanRL.Networks: hand it over.)
CleanRLHip.ppo(config; make_logger = CleanRL.Logger.make_logger,
               init = (n_act, obs_dim, hidden) -> CleanRLHip.reference_params(CleanRL.Networks, CleanRL.Flux, n_act, obs_dim, hidden))
