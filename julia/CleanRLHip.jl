# CleanRLHip.jl — the Julia shell a CleanRL.jl maintainer drops next to src/algorithms/ppo.jl.
#
# It keeps the reference's surface (PPOConfig, ppo, get_action, logprob_actions, gae and the two @info records of
# src/algorithms/ppo.jl:157,247) and forwards every numeric step to libcleanrl_hip.so through `ccall`.
# NOT EXECUTED IN THIS REPO'S CI: the build image has no Julia (SURVEY.md F4). The same C entry points are exercised
# through the ctypes mirror (cleanrl.jl_amd/_lib.py) by tests/; the struct layout below is the one
# tests/test_host_cpu.py::test_config_struct_matches_header_and_reference_defaults pins (104 bytes).
module CleanRLHip

export PPOConfig, ppo, get_action, logprob_actions, gae, a2c, dqn, q_values, reference_params, init_params!, comm_unique_id, comm_init!, comm_peer_export!, comm_peer_attach!,
       comm_destroy!, set_option!, get_option

const libcrl = get(ENV, "CLEANRL_HIP_LIB", joinpath(@__DIR__, "..", "cleanrl.jl_amd", "libcleanrl_hip.so"))

# ppo.jl:1-19 — unchanged
Base.@kwdef struct PPOConfig
  total_timesteps::Int = 500_000
  num_steps::Int = 32
  num_envs::Int = 4
  num_minibatches::Int = 4
  update_epochs::Int = 4
  lr::Float32 = 2.5f-4
  gamma::Float32 = 0.99
  gae_lambda::Float32 = 0.95
  clip_coef::Float32 = 0.2
  ent_coeff::Float32 = 0.01
  v_coef::Float32 = 0.5
  normalize_advantages::Bool = true
  clip_value_loss::Bool = true
  anneal_lr::Bool = true
end

# crl_ppo_config (include/cleanrl_hip.h) — isbits, C layout
struct CrlConfig
  total_timesteps::Int64
  num_steps::Int32; num_envs::Int32; num_minibatches::Int32; update_epochs::Int32
  lr::Float32; gamma::Float32; gae_lambda::Float32; clip_coef::Float32; ent_coeff::Float32; v_coef::Float32
  normalize_advantages::Int32; clip_value_loss::Int32; anneal_lr::Int32
  obs_dim::Int32; n_act::Int32; hidden::Int32; gae_mode::Int32; env_kind::Int32; stale_obs::Int32
  env_id_offset::Int32; shuffle_mode::Int32
  seed::UInt64
end

struct CrlStats            # crl_ppo_stats
  loss::Float64; pg_loss::Float64; v_loss::Float64; entropy_loss::Float64
  adv_mean::Float64; adv_std::Float64; u_value::Float64; n_unclipped_wins::Float64
end
struct CrlEpisodeStats     # crl_episode_stats
  episodes::Float64; return_sum::Float64; length_sum::Float64; return_max::Float64
end
struct CrlEpisodeRecord    # crl_episode_record: one per finished episode (ppo.jl:147-162)
  episode_return::Float32; episode_length::Int32; env::Int32; step::Int32
end
struct CrlIterationReport  # crl_ppo_iteration_report: whose records crl_ppo_iterate_async / crl_ppo_drain handed back
  iteration::Int64; episodes::CrlEpisodeStats; n_episodes::Int64; n_ring::Int32; pad::Int32
end

check(rc::Int32) = rc == 0 || error(unsafe_string(ccall((:crl_last_error, libcrl), Cstring, ())))

mutable struct Agent
  h::Ptr{Cvoid}
  config::PPOConfig
  obs_dim::Int; n_act::Int; hidden::Int; device::Int
  # The reference derives the shapes from the env and the network builder (ppo.jl:85-87: single_state_space / single_action_space;
  # networks.jl:36-38: make_actor_critic(action_space, obs_space, hidden_sizes = [64, 64])); here they are keywords with the same
  # defaults — obs_dim = length(single_state_space), n_act = length(single_action_space), hidden = hidden_sizes[1] (both hidden layers
  # are equal, as in the reference). 4 / 2 / 64 runs the fused kernels; any other shape (obs ≤ 64, act ≤ 16, hidden 64 / 128 / 256 —
  # e.g. the LunarLander-shaped 8 / 4 / 256 of BASELINE config 3) runs the layer-wise path and needs env_kind 1 (synthetic) or 2 (external).
  # gae_mode 0 = the reference's loop (ppo.jl:66), 1 = bootstrap from next_value; stale_obs = the reference's Q7 behaviour.
  # shuffle_mode 2 = CRL_SHUFFLE_BLOCKED_FY: an exact uniform shuffle like Random.shuffle (ppo.jl:194); same default as the
  # ctypes mirror (cleanrl.jl_amd/ppo.py). 0 = serial Fisher–Yates, 1 = keyed bijection (pseudo-random, faster).
  function Agent(config::PPOConfig; obs_dim::Integer=4, n_act::Integer=2, hidden::Integer=64, env_kind::Integer=(obs_dim == 4 && n_act == 2 ? 0 : 1),
                 gae_mode::Integer=0, stale_obs::Bool=true, device::Integer=0, seed=0x5EED, env_id_offset::Integer=0, shuffle_mode::Integer=2)
    c = CrlConfig(config.total_timesteps, config.num_steps, config.num_envs, config.num_minibatches, config.update_epochs,
                  config.lr, config.gamma, config.gae_lambda, config.clip_coef, config.ent_coeff, config.v_coef,
                  config.normalize_advantages, config.clip_value_loss, config.anneal_lr,
                  obs_dim, n_act, hidden, gae_mode, env_kind, stale_obs, env_id_offset, shuffle_mode, seed)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:crl_ppo_create, libcrl), Int32, (Ref{CrlConfig}, Int32, Ref{Ptr{Cvoid}}), c, device, out))
    a = new(out[], config, obs_dim, n_act, hidden, device)
    finalizer(x -> ccall((:crl_ppo_destroy, libcrl), Int32, (Ptr{Cvoid},), x.h), a)
  end
end
# number of Float32 parameters of Flux.params(actor, critic) for the handle's shapes (networks.jl:36-49)
function param_count(a::Agent)
  n = Ref{Int64}(0)
  check(ccall((:crl_ppo_param_count, libcrl), Int32, (Ptr{Cvoid}, Ref{Int64}), a.h, n))
  Int(n[])
end

# Flux.params(actor, critic) → one flat Float32 vector in the same order (ppo.jl:196); W is (out,in) column-major
# exactly as Flux stores it, so `vcat(vec.(Flux.params(actor, critic))...)` is the argument.
set_params!(a::Agent, flat::Vector{Float32}) =
  GC.@preserve flat check(ccall((:crl_ppo_write, libcrl), Int32, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Csize_t), a.h, 9, pointer(flat), sizeof(flat)))

# ppo.jl:87 — `actor, critic = Networks.make_actor_critic(single_act_space, single_obs_space) .|> Flux.f32` — when the caller has no Flux
# networks to upload: the library's own host-side restatement of networks.jl:36-49 (orthogonal weights, gains √2 / 0.01 / 1.0, zero biases;
# crl_ppo_init_params). A fresh handle holds zeros and every computing entry point refuses to run on them ("parameters not set").
init_params!(a::Agent, seed::Integer=0) = check(ccall((:crl_ppo_init_params, libcrl), Int32, (Ptr{Cvoid}, UInt64), a.h, UInt64(seed)))

# The reference's own builder, flattened the way the boundary wants it: `Networks` is the reference's module (src/utils/networks.jl), `Flux` the
# Flux it loaded. make_actor_critic only takes `length` of its two spaces (networks.jl:37-38), so Base.OneTo stands in for them.
function reference_params(Networks, Flux, n_act::Integer, obs_dim::Integer, hidden::Integer)
  actor, critic = Networks.make_actor_critic(Base.OneTo(n_act), Base.OneTo(obs_dim), Int[hidden, hidden]) .|> Flux.f32     # ppo.jl:87
  Vector{Float32}(vcat(vec.(Flux.params(actor, critic))...))                                                             # ppo.jl:196 order
end
# Default network builder of ppo / a2c: inside the reference package (this file included next to src/algorithms/ppo.jl, where `Networks` and
# `Flux` are names of the enclosing module, src/CleanRL.jl:12,23) the reference's own make_actor_critic; standalone `nothing`, and the entry
# points then take the library's initialiser — never zeros.
function _default_init()
  pm = parentmodule(@__MODULE__)
  (isdefined(pm, :Networks) && isdefined(pm, :Flux)) || return nothing
  (n_act, obs_dim, hidden) -> reference_params(getfield(pm, :Networks), getfield(pm, :Flux), n_act, obs_dim, hidden)
end

# ppo.jl:21-32 — `actor` is the Agent holding the weights on the GPU; u are the rand() draws of StatsBase.sample
function get_action(obs::AbstractVecOrMat{Float32}, actor::Agent; u::Vector{Float64}=rand(size(obs, ndims(obs))))
  n = size(obs, ndims(obs)); o = Array(obs)
  action = Vector{Int32}(undef, n); logprob = Vector{Float32}(undef, n)
  GC.@preserve o u action logprob check(ccall((:crl_policy_act, libcrl), Int32,
    (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float64}, Int32, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}),
    actor.h, o, u, n, action, logprob, C_NULL))
  Int.(action) .+ 1, logprob                      # 1-based like Base.OneTo(2) (ppo.jl:26)
end

# ppo.jl:34-45
function logprob_actions(obs::AbstractVecOrMat{Float32}, actor::Agent, actions::AbstractVector{Int32})
  n = size(obs, ndims(obs)); o = Array(obs); a0 = Int32.(actions .- 1)
  logprob = Vector{Float32}(undef, n); entropy = Matrix{Float32}(undef, actor.n_act, n)   # (n_act, batch): -p .* logp (ppo.jl:42)
  GC.@preserve o a0 logprob entropy check(ccall((:crl_logprob_actions, libcrl), Int32,
    (Ptr{Cvoid}, Ptr{Float32}, Ptr{Int32}, Int32, Ptr{Float32}, Ptr{Float32}), actor.h, o, a0, n, logprob, entropy))
  logprob, entropy
end

# ppo.jl:48-73 — one env; values [0,k], rewards [1,k], terminals [0,k]. The last slot is 0 (upstream: uninitialised).
# `device` names the GPU the stateless scan runs on (an Agent's own is agent.device).
function gae(values::AbstractVector{Float32}, rewards::AbstractVector{Float32}, terminals::AbstractVector{Bool},
             γ::Float32, λ::Float32; mode::Integer=0, device::Integer=0, seg::Integer=0, tile::Integer=0, nt_loads::Integer=2)
  k = length(rewards)
  v = Vector{Float32}(values[1:k]); r = Vector{Float32}(rewards); t = UInt8.(terminals[1:k])   # BitArray → bytes
  nv = Float32[values[k+1]]; nd = UInt8[terminals[k+1]]
  adv = Vector{Float32}(undef, k)
  GC.@preserve v r t nv nd adv check(ccall((:crl_gae_opt, libcrl), Int32,
    (Int32, Ptr{Float32}, Ptr{Float32}, Ptr{UInt8}, Ptr{Float32}, Ptr{UInt8}, Int32, Int32, Float32, Float32, Int32,
     Ptr{Float32}, Ptr{Float32}, Int32, Int32, Int32), device, v, r, t, nv, nd, 1, k, γ, λ, mode, adv, C_NULL, seg, tile, nt_loads))
  adv
end

# Data parallelism over num_envs, one Julia process per GPU (the reference is single-process): rank 0 draws the 128-byte RCCL
# id, the launcher (MPI.jl, Distributed, a file) hands it to every rank, each rank attaches its shard handle.
function comm_unique_id()
  id = Vector{UInt8}(undef, 128)
  GC.@preserve id check(ccall((:crl_comm_unique_id, libcrl), Int32, (Ptr{UInt8},), id))
  id
end
comm_init!(a::Agent, id::Vector{UInt8}, world_size::Integer, rank::Integer) =
  GC.@preserve id check(ccall((:crl_comm_init, libcrl), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32), a.h, id, world_size, rank))

# The same exchange without RCCL: a one-shot all-reduce over peer-mapped mailboxes (csrc/peer.hip). Every rank exports its
# mailbox (64-byte hipIpcMemHandle_t), the launcher all-gathers the handles, every rank attaches all of them in rank order.
function comm_peer_export!(a::Agent, world_size::Integer, rank::Integer)
  hnd = Vector{UInt8}(undef, 64)
  GC.@preserve hnd check(ccall((:crl_comm_peer_export, libcrl), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), a.h, world_size, rank, hnd))
  hnd
end
comm_peer_attach!(a::Agent, handles::Vector{UInt8}) =
  GC.@preserve handles check(ccall((:crl_comm_peer_attach, libcrl), Int32, (Ptr{Cvoid}, Ptr{UInt8}), a.h, handles))

# Detaches whatever exchange is attached (a launcher falling back from a partially failed RCCL start to the peer all-reduce)
comm_destroy!(a::Agent) = check(ccall((:crl_comm_destroy, libcrl), Int32, (Ptr{Cvoid},), a.h))

# Per-handle options (include/cleanrl_hip.h lists them): kernel-flavour / numerics switches, integer-valued, by name, e.g.
# set_option!(agent, "gemm", 1) selects the bf16x3 flavour, set_option!(agent, "guard_window", 1) a read-back per iteration.
set_option!(a::Agent, key::AbstractString, value::Integer) =
  check(ccall((:crl_ppo_set_option, libcrl), Int32, (Ptr{Cvoid}, Cstring, Int64), a.h, key, value))
function get_option(a::Agent, key::AbstractString)
  v = Ref{Int64}(0)
  check(ccall((:crl_ppo_get_option, libcrl), Int32, (Ptr{Cvoid}, Cstring, Ref{Int64}), a.h, key, v))
  v[]
end

# ppo.jl:75 — same signature; the loop body (ppo.jl:117-253) runs on the GPU, one ccall per update.
# episode_records > 0 turns on the device ring (crl_episode_ring_enable): every finished episode leaves {return, length, env,
# step} and `ppo` emits ONE "Episode Statistics" record per episode in the reference's order (ppo.jl:147-165: step by step,
# done envs ascending; global_step as of that step, ppo.jl:124). With 0 it emits one aggregate record per update (65536 envs
# finish ~10^5 episodes per rollout). Multi-GPU: pass comm = (id, world_size, rank) and a config whose num_envs is the shard.
# Shapes other than the reference's CartPole 4 / 2 / 64 are keywords forwarded to Agent (obs_dim, n_act, hidden, env_kind, gae_mode,
# stale_obs): ppo(cfg; obs_dim = 8, n_act = 4, hidden = 256) is BASELINE config 3 on the synthetic env.
# ppo.jl:77 installs the global logger before anything else (`Logger.make_logger("ppo-2-test"; to_terminal=false)`, logger.jl:7-29): `make_logger`
# is that function — by default the `Logger` module of the package this file is included into (src/CleanRL.jl includes utils/logger.jl before
# the algorithms), `nothing` when there is none (standalone use: the caller's own global logger receives the @info records).
_default_make_logger() = isdefined(parentmodule(@__MODULE__), :Logger) ? getfield(parentmodule(@__MODULE__), :Logger).make_logger : nothing
# ppo.jl:87 — the networks: `params` (a caller's own `vcat(vec.(Flux.params(actor, critic))...)`) wins; otherwise `init(n_act, obs_dim, hidden)` builds them —
# by default the reference's Networks.make_actor_critic(...) .|> Flux.f32 when this file lives inside the reference package (_default_init), and
# the library's restatement of it (init_params!, seeded by init_seed) when it does not. There is no path that leaves the handle's zeros in place:
# under data parallelism every rank must end up with the SAME parameters, so pass `params` or a seeded init there (init_params! with one
# init_seed is identical on all ranks; the reference builder draws from each process's own RNG).
function ppo(config::PPOConfig=PPOConfig(); device::Integer=0, params::Union{Nothing,Vector{Float32}}=nothing, init=_default_init(), init_seed::Integer=0,
             episode_records::Integer=4096, comm::Union{Nothing,Tuple{Vector{UInt8},Int,Int}}=nothing, run_name::AbstractString="ppo-2-test",
             make_logger=_default_make_logger(), shape...)
  make_logger === nothing || make_logger(run_name; to_terminal=false)      # ppo.jl:77
  world, rank = comm === nothing ? (1, 0) : (comm[2], comm[3])
  agent = Agent(config; device, env_id_offset=rank * config.num_envs, shape...)
  if params === nothing && init !== nothing && world == 1
    params = init(agent.n_act, agent.obs_dim, agent.hidden)                # ppo.jl:87 with the reference's own builder
  end
  if params === nothing
    init_params!(agent, init_seed)                                         # ppo.jl:87 through crl_ppo_init_params (same on every rank)
  else
    length(params) == param_count(agent) || error("params has $(length(params)) entries, the networks have $(param_count(agent))")
    set_params!(agent, params)
  end
  comm === nothing || comm_init!(agent, comm[1], world, rank)
  episode_records > 0 && check(ccall((:crl_episode_ring_enable, libcrl), Int32, (Ptr{Cvoid}, Int32), agent.h, episode_records))
  check(ccall((:crl_env_reset, libcrl), Int32, (Ptr{Cvoid},), agent.h))
  batch_size = config.num_steps * config.num_envs * world          # ppo.jl:89 over the whole job
  num_updates = config.total_timesteps ÷ batch_size                # ppo.jl:91
  nstats = config.update_epochs * config.num_minibatches
  stats = Vector{CrlStats}(undef, nstats)
  recs = Vector{CrlEpisodeRecord}(undef, max(episode_records, 1)); rep = Ref{CrlIterationReport}()
  last_log_step = 0; start_time = time()
  # the records of one update, in the reference's order: its episodes (ppo.jl:147-165), then its minibatches (ppo.jl:246-248)
  function emit(r::CrlIterationReport)
    base = r.iteration * batch_size
    global_step = base + batch_size
    if episode_records > 0
      for e in sort!(recs[1:r.n_ring]; by = x -> (x.step, x.env))          # the reference's order: step by step, done envs ascending
        gs = base + (e.step + 1) * config.num_envs * world                # ppo.jl:124 global_step += num_envs per step
        steps_per_sec = trunc(gs / (time() - start_time))
        log_step_inc = last_log_step == 0 ? 0 : gs - last_log_step
        @info "Episode Statistics" episode_return = e.episode_return episode_length = e.episode_length global_step = gs steps_per_sec log_step_increment = log_step_inc
        last_log_step = gs
      end
    elseif r.episodes.episodes > 0
      steps_per_sec = trunc(global_step / (time() - start_time))
      episode_return = r.episodes.return_sum / r.episodes.episodes; episode_length = r.episodes.length_sum / r.episodes.episodes
      log_step_inc = last_log_step == 0 ? 0 : global_step - last_log_step
      @info "Episode Statistics" episode_return episode_length global_step steps_per_sec log_step_increment = log_step_inc
      last_log_step = global_step
    end
    for s in stats
      log_step_inc = last_log_step == 0 ? 0 : global_step - last_log_step
      @info "Training Statistics" loss = s.loss pg_loss = s.pg_loss v_loss = s.v_loss entropy_loss = s.entropy_loss log_step_increment = log_step_inc
      last_log_step = global_step
    end
  end
  # Pipelined read-back (crl_ppo_iterate_async): update k's records are handed over after update k + 1 has been enqueued, so the GPU never waits while Julia
  # logs; the record stream is the reference's, one update late, and crl_ppo_drain hands over the last one.
  for update in 1:num_updates
    GC.@preserve stats recs check(ccall((:crl_ppo_iterate_async, libcrl), Int32, (Ptr{Cvoid}, Ref{CrlIterationReport}, Ptr{CrlStats}, Ptr{CrlEpisodeRecord}, Int32),
                                        agent.h, rep, stats, recs, episode_records))
    rep[].iteration >= 0 && emit(rep[])
  end
  GC.@preserve stats recs check(ccall((:crl_ppo_drain, libcrl), Int32, (Ptr{Cvoid}, Ref{CrlIterationReport}, Ptr{CrlStats}, Ptr{CrlEpisodeRecord}, Int32),
                                      agent.h, rep, stats, recs, episode_records))
  rep[].iteration >= 0 && emit(rep[])
  agent
end

# ------------------------------------------------------------------------------------------------------
# a2c.jl — same names: A2CConfig keeps the reference's fields (a2c.jl:1-10); the loop body runs on the GPU
# ------------------------------------------------------------------------------------------------------
struct CrlA2CConfig       # include/cleanrl_hip.h crl_a2c_config
  lr::Float64; total_timesteps::Int64; min_replay_size::Int32; max_steps::Int32; gamma::Float64; seed::UInt64
end
struct CrlA2CTrainStats; actor_loss::Float64; critic_loss::Float64; n::Int32; trained::Int32; end
struct CrlA2CEpisode; episode_return::Float64; episode_length::Int64; global_step::Int64; end

# a2c.jl:13-24 — same signature
function discounted_future_rewards(rewards::Vector{Float64}, terminals::Vector{Bool}, final_value::Float64, γ::Float64; device=0)
  out = similar(rewards); t = UInt8.(terminals)
  GC.@preserve rewards t out check(ccall((:crl_a2c_discounted_future_rewards, libcrl), Int32,
    (Int32, Ptr{Float64}, Ptr{UInt8}, Int32, Float64, Float64, Ptr{Float64}), device, rewards, t, length(rewards), final_value, γ, out))
  out
end

# a2c.jl:29 — same signature; `config` is the reference's A2CConfig. a2c.jl:30 installs the logger (terminal + TensorBoard: make_logger's own
# defaults), a2c.jl:37 builds the networks: `params` wins, else `init` (the reference's Networks.make_actor_critic inside the package — a2c.jl:37
# does not pipe through Flux.f32, Flux.orthogonal is Float32 already), else the library's initialiser. Never the handle's zeros.
function a2c(config; device=0, seed=UInt64(0x5EED), params::Union{Nothing,Vector{Float32}}=nothing, init=_default_init(), init_seed::Integer=0,
             make_logger=_default_make_logger())
  make_logger === nothing || make_logger("a2c|$(config.run_name)")         # a2c.jl:30
  cfg = CrlA2CConfig(config.lr, config.total_timesteps, config.min_replay_size, 500, config.gamma, seed)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:crl_a2c_create, libcrl), Int32, (Ref{CrlA2CConfig}, Int32, Ref{Ptr{Cvoid}}), cfg, device, h))
  params === nothing && init !== nothing && (params = init(2, 4, 64))      # a2c.jl:37 make_actor_critic(env): CartPole 4 / 2, hidden [64, 64]
  if params === nothing
    check(ccall((:crl_a2c_init_params, libcrl), Int32, (Ptr{Cvoid}, UInt64), h[], UInt64(init_seed)))
  else
    GC.@preserve params check(ccall((:crl_a2c_write_params, libcrl), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), h[], params, length(params)))
  end
  stats = Ref{CrlA2CTrainStats}(); eps = Vector{CrlA2CEpisode}(undef, 4096); n_eps = Ref{Int32}(0); taken = Ref{Int64}(0)
  start_time = time(); global_step = 0
  while global_step < config.total_timesteps
    GC.@preserve eps check(ccall((:crl_a2c_run_until_update, libcrl), Int32,
      (Ptr{Cvoid}, Int64, Ref{CrlA2CTrainStats}, Ptr{CrlA2CEpisode}, Int32, Ref{Int32}, Ref{Int64}),
      h[], typemax(Int64), stats, eps, length(eps), n_eps, taken))
    taken[] == 0 && break
    global_step += taken[]
    # the episode whose end triggered the update is the call's LAST record, and the reference logs the update first (a2c.jl:100, then :106)
    for (i, e) in enumerate(@view eps[1:n_eps[]])
      i == n_eps[] && stats[].trained == 1 && @info "Training Statistics" actor_loss = stats[].actor_loss critic_loss = stats[].critic_loss
      steps_per_sec = trunc(e.global_step / (time() - start_time))
      @info "Episode Statistics" episode_return = e.episode_return episode_length = e.episode_length global_step = e.global_step steps_per_sec
    end
    n_eps[] == 0 && stats[].trained == 1 && @info "Training Statistics" actor_loss = stats[].actor_loss critic_loss = stats[].critic_loss
  end
  check(ccall((:crl_a2c_destroy, libcrl), Int32, (Ptr{Cvoid},), h[]))
end

# ------------------------------------------------------------------------------------------------------
# dqn.jl — same names: DQNConfig keeps the reference's fields (dqn.jl:1-19); the loop body runs on the GPU
# ------------------------------------------------------------------------------------------------------
struct CrlDQNConfig       # include/cleanrl_hip.h crl_dqn_config
  log_frequency::Int64; total_timesteps::Int64; buffer_size::Int64; min_buff_size::Int64
  lr::Float64
  train_freq::Int64; target_net_freq::Int64; batch_size::Int64
  gamma::Float64; epsilon_start::Float64; epsilon_end::Float64; epsilon_duration::Float64
  max_steps::Int32; pad::Int32
  seed::UInt64
end
struct CrlDQNEpisode; episode_return::Float64; episode_length::Int64; global_step::Int64; epsilon::Float64; end   # dqn.jl:88
struct CrlDQNLossRecord; global_step::Int64; loss::Float64; end                                                    # dqn.jl:116

mutable struct DQNAgent
  h::Ptr{Cvoid}
end

# q_net(obs) (dqn.jl:64) for observations (4, n) Float64 → (2, n) Float64
function q_values(agent::DQNAgent, obs::AbstractMatrix{Float64})
  o = Array(obs); n = size(o, 2); q = Matrix{Float64}(undef, 2, n)
  GC.@preserve o q check(ccall((:crl_dqn_q_values, libcrl), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}), agent.h, o, n, q))
  q
end

# dqn.jl:34 — same signature; `config` is the reference's DQNConfig (note its field `log_frequencey`, sic), `params` =
# vcat(vec.(Flux.params(q_net))...) of make_nn(env) (dqn.jl:22-26: 4 → 120 → 84 → 2, 10,934 parameters)
# dqn.jl:35 installs the logger, dqn.jl:39 builds q_net: `params` wins, else `init()` (inside the reference package its own make_nn(CartPoleEnv()),
# dqn.jl:22-26), else the library's initialiser (crl_dqn_init_params: the same glorot-uniform layers). Never the handle's zeros.
function _default_dqn_init()
  pm = parentmodule(@__MODULE__)
  (isdefined(pm, :make_nn) && isdefined(pm, :CartPoleEnv) && isdefined(pm, :Flux)) || return nothing
  () -> Vector{Float32}(vcat(vec.(getfield(pm, :Flux).params(getfield(pm, :make_nn)(getfield(pm, :CartPoleEnv)())))...))
end
function dqn(config; device=0, seed=UInt64(0x5EED), params::Union{Nothing,Vector{Float32}}=nothing, init=_default_dqn_init(), init_seed::Integer=0,
             chunk::Integer=10_000, make_logger=_default_make_logger())
  make_logger === nothing || make_logger("dqn|$(config.run_name)")         # dqn.jl:35
  cfg = CrlDQNConfig(config.log_frequencey, config.total_timesteps, config.buffer_size, config.min_buff_size, config.lr,
                     config.train_freq, config.target_net_freq, config.batch_size, config.gamma, config.epsilon_start,
                     config.epsilon_end, config.epsilon_duration, 200, 0, seed)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:crl_dqn_create, libcrl), Int32, (Ref{CrlDQNConfig}, Int32, Ref{Ptr{Cvoid}}), cfg, device, h))
  agent = DQNAgent(h[])
  finalizer(x -> ccall((:crl_dqn_destroy, libcrl), Int32, (Ptr{Cvoid},), x.h), agent)
  params === nothing && init !== nothing && (params = init())             # dqn.jl:39 make_nn(env)
  if params === nothing
    check(ccall((:crl_dqn_init_params, libcrl), Int32, (Ptr{Cvoid}, UInt64), agent.h, UInt64(init_seed)))
  else
    GC.@preserve params check(ccall((:crl_dqn_write_params, libcrl), Int32, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), agent.h, params, length(params)))
  end
  eps = Vector{CrlDQNEpisode}(undef, 8192); losses = Vector{CrlDQNLossRecord}(undef, 4096)
  n_eps = Ref{Int32}(0); n_losses = Ref{Int32}(0); taken = Ref{Int64}(0)
  start_time = time(); global_step = 0
  while global_step < config.total_timesteps
    GC.@preserve eps losses check(ccall((:crl_dqn_run, libcrl), Int32,
      (Ptr{Cvoid}, Int64, Ptr{CrlDQNEpisode}, Int32, Ref{Int32}, Ptr{CrlDQNLossRecord}, Int32, Ref{Int32}, Ref{Int64}),
      agent.h, chunk, eps, length(eps), n_eps, losses, length(losses), n_losses, taken))
    taken[] == 0 && break
    global_step += taken[]
    for e in @view eps[1:n_eps[]]
      steps_per_sec = trunc(e.global_step / (time() - start_time))
      @info "Episode Statistics" episode_return = e.episode_return episode_length = e.episode_length global_step = e.global_step steps_per_sec ϵ = e.epsilon
    end
    for l in @view losses[1:n_losses[]]
      @info "Training Statistics" loss = l.loss global_step = l.global_step
    end
  end
  agent
end

end # module
