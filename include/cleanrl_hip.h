/*
 * cleanrl_hip.h — C ABI of libcleanrl_hip.so: the MI355X-native PPO rollout + GAE + update hot path.
 *
 * Drop-in boundary for sash-a/CleanRL.jl `src/algorithms/ppo.jl`. The reference has no FFI today (it is pure
 * Julia); each entry point below names the reference function / lines it replaces, and INTEGRATION.md shows the
 * `ccall` a maintainer adds on the Julia side. Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *  - every call returns 0 on success, non-zero on error; crl_last_error() gives the message (thread-local).
 *    No exception crosses the boundary. Julia wrapper: `rc == 0 || error(unsafe_string(crl_last_error()))`.
 *  - host pointers are borrowed for the duration of the call only (Julia: GC.@preserve); device memory is owned by
 *    the library behind the opaque crl_ppo handle.
 *  - layouts are Julia column-major, passed without copies: obs (obs_dim, n); buffers (nt, k); flat sample index
 *    b = e + nt*t (ppo.jl:184-189). Actions are 0-based int32 here (Julia side adds 1); terminals are uint8
 *    (widen the env's BitArray to Vector{Bool} first, multi_thread_env.jl:18).
 *  - parameters travel as ONE flat float vector in Flux.params(actor, critic) order (ppo.jl:196):
 *    actor W1(h,obs) b1(h) W2(h,h) b2(h) W3(A,h) b3(A), critic W1 b1 W2 b2 W3(1,h) b3(1); W is (out,in) col-major.
 *  - calls on one handle must be serialised by the caller; work is asynchronous on the handle's HIP stream until
 *    crl_sync() or a call that copies results to the host.
 */
#ifndef CLEANRL_HIP_H
#define CLEANRL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRL_VERSION 100 /* 0.1.0 */

/* gae_mode */
#define CRL_GAE_COMPAT 0 /* ppo.jl:66 loop k-1:-1:1, carry 0, slot k defined as 0 (upstream leaves it uninitialised) */
#define CRL_GAE_FIXED 1  /* loop from k with the bootstrap value (CleanRL-python semantics) */
/* env_kind */
#define CRL_ENV_CARTPOLE 0 /* CartPoleEnv(T=Float32, max_steps=500) ppo.jl:82 */
#define CRL_ENV_SYNTHETIC 1 /* stateless generator for shapes the reference has no env for: obs ~ U(-1,1)^d, reward ~ U(-1,1), done ~ B(1/200) */
#define CRL_ENV_EXTERNAL 2 /* envs stepped by the caller: crl_policy_act + crl_rollout_store */
/* shuffle mode */
#define CRL_SHUFFLE_FISHER_YATES 0 /* exact serial Fisher–Yates on device (ppo.jl:194 semantics) */
#define CRL_SHUFFLE_BIJECTION 1     /* perm[p] = keyed bijection of [0,B): O(1) per element, pseudo-random (throughput path) */
#define CRL_SHUFFLE_BLOCKED_FY 2    /* exact parallel shuffle: random split into ~64-element buckets + Fisher–Yates in each */
/* rollout / state fields for crl_ppo_read / crl_ppo_write */
enum crl_field {
  CRL_F_OBS = 0,      /* float  (obs_dim, nt, k)  replay_buffer.jl:15-18 / ppo.jl:96 */
  CRL_F_ACTION = 1,   /* int32  (nt, k) 0-based   ppo.jl:97  */
  CRL_F_LOGPROB = 2,  /* float  (nt, k)           ppo.jl:98  */
  CRL_F_REWARD = 3,   /* float  (nt, k)           ppo.jl:99  */
  CRL_F_TERMINAL = 4, /* uint8  (nt, k)           ppo.jl:100 */
  CRL_F_VALUE = 5,    /* float  (nt, k)           ppo.jl:101 */
  CRL_F_ADVANTAGE = 6,/* float  (nt, k)           ppo.jl:180 */
  CRL_F_RETURN = 7,   /* float  (nt, k)           ppo.jl:181 */
  CRL_F_PERM = 8,     /* int32  (nt*k) 0-based    ppo.jl:194 b_inds */
  CRL_F_PARAMS = 9,   /* float  (P)               ppo.jl:196 */
  CRL_F_GRADS = 10,   /* float  (P) gradient of the last minibatch (after all-reduce, before ClipNorm) ppo.jl:202 */
  CRL_F_ADAM_M = 11,  /* float  (P) */
  CRL_F_ADAM_V = 12,  /* float  (P) */
  CRL_F_ENV_STATE = 13, /* float (obs_dim, nt) env-internal state */
  CRL_F_CUR_OBS = 14,   /* float (obs_dim, nt) next_obs of ppo.jl:114,143 */
  CRL_F_NEXT_DONE = 15, /* uint8 (nt)          next_done of ppo.jl:115,144 */
  CRL_F_ENV_T = 16,     /* int32 (nt) steps since reset */
  CRL_F_BETAP = 17,     /* double (24) Adam running beta powers per array */
  CRL_F_ADV_SUMS = 18   /* double (num_minibatches, 2) Σadv, Σadv² of the current permutation slices (what the ranks all-reduce) */
};

/* Mirror of PPOConfig (ppo.jl:1-19) + the shapes the reference hard-codes (networks.jl:36, CartPole) */
typedef struct crl_ppo_config {
  int64_t total_timesteps;      /* ppo.jl:2 */
  int32_t num_steps;            /* ppo.jl:3 */
  int32_t num_envs;             /* ppo.jl:4  — envs owned by THIS handle (one data-parallel shard) */
  int32_t num_minibatches;      /* ppo.jl:5 */
  int32_t update_epochs;        /* ppo.jl:6 */
  float lr;                     /* ppo.jl:8 */
  float gamma;                  /* ppo.jl:9 */
  float gae_lambda;             /* ppo.jl:10 */
  float clip_coef;              /* ppo.jl:12 */
  float ent_coeff;              /* ppo.jl:13 */
  float v_coef;                 /* ppo.jl:14 */
  int32_t normalize_advantages; /* ppo.jl:16 — must be 1 (0 is an error upstream too, ppo.jl:219-222) */
  int32_t clip_value_loss;      /* ppo.jl:17 */
  int32_t anneal_lr;            /* ppo.jl:18 */
  int32_t obs_dim;              /* 4 */
  int32_t n_act;                /* 2 */
  int32_t hidden;               /* 64 (networks.jl:36) */
  int32_t gae_mode;             /* CRL_GAE_* */
  int32_t env_kind;             /* CRL_ENV_* */
  int32_t stale_obs;            /* 1 = ppo.jl:143-164 behaviour: policy sees the terminal obs after a reset */
  int32_t env_id_offset;        /* global id of local env 0 (rank * num_envs under data parallelism) */
  int32_t shuffle_mode;         /* CRL_SHUFFLE_* */
  uint64_t seed;
} crl_ppo_config;

/* "Training Statistics" record of ppo.jl:247, one per optimiser step */
typedef struct crl_ppo_stats {
  double loss, pg_loss, v_loss, entropy_loss;
  double adv_mean, adv_std, u_value, n_unclipped_wins;
} crl_ppo_stats;

/* "Episode Statistics" of ppo.jl:147-162 aggregated over one rollout */
typedef struct crl_episode_stats {
  double episodes, return_sum, length_sum, return_max;
} crl_episode_stats;

typedef struct crl_ppo crl_ppo;

int32_t crl_version(void);
const char* crl_last_error(void);
int32_t crl_device_count(int32_t* n);

/* ppo.jl:75-115 setup: buffers, optimiser state, env state — all resident in HBM on `device`. */
int32_t crl_ppo_create(const crl_ppo_config* cfg, int32_t device, crl_ppo** out);
int32_t crl_ppo_destroy(crl_ppo* h);
int32_t crl_ppo_param_count(const crl_ppo* h, int64_t* n);
int32_t crl_sync(crl_ppo* h);

/* `actor, critic = Networks.make_actor_critic(single_act_space, single_obs_space) .|> Flux.f32` — ppo.jl:87 / networks.jl:36-49: separate
 * actor and critic Dense(in,h,tanh_fast) → Dense(h,h,tanh_fast) → Dense(h,out), Flux.orthogonal weights with gains √2 / √2 / 0.01 (actor
 * head) and √2 / √2 / 1.0 (critic head), zero biases, as one flat float vector in Flux.params(actor, critic) order. Stateless, host-only
 * (runs without a GPU). The random stream is the library's own (seeded; Flux's task-local Xoshiro stream is not reproducible outside
 * Julia) — the distribution is the reference's. n must be the parameter count of the shape (crl_ppo_param_count). */
int32_t crl_make_actor_critic(int32_t obs_dim, int32_t n_act, int32_t hidden, uint64_t seed, float* out, size_t n);
/* ppo.jl:87 for a handle: crl_make_actor_critic for its shape, uploaded like a CRL_F_PARAMS write. A FRESH HANDLE HOLDS ALL-ZERO
 * PARAMETERS (h1 = h2 = 0: every gradient but the head biases' is zero for ever), so every entry point that computes with the networks —
 * crl_policy_act, crl_logprob_actions, crl_rollout_run, crl_ppo_update_minibatch, crl_ppo_iterate, crl_compute_gae in fixed mode —
 * returns an error ("parameters not set") until CRL_F_PARAMS has been written or this call has run. */
int32_t crl_ppo_init_params(crl_ppo* h, uint64_t seed);

/* Raw field access (parity tests, checkpointing, Julia-side inspection). nbytes must match the field size. */
int32_t crl_ppo_write(crl_ppo* h, int32_t field, const void* host, size_t nbytes);
int32_t crl_ppo_read(crl_ppo* h, int32_t field, void* host, size_t nbytes);

/* get_action(obs, actor) + critic(obs) — ppo.jl:21-32,127-128. obs (obs_dim,n) f32, u[n] uniform f64 in [0,1)
 * (the rand() of StatsBase.sample); outputs action[n] (0-based), logprob[n], value[n] (may be NULL). */
int32_t crl_policy_act(crl_ppo* h, const float* obs, const double* u, int32_t n, int32_t* action, float* logprob,
                       float* value);
/* logprob_actions(obs, actor, actions) — ppo.jl:34-45. entropy is the (n_act,n) matrix -p.*logp. */
int32_t crl_logprob_actions(crl_ppo* h, const float* obs, const int32_t* actions, int32_t n, float* logprob,
                            float* entropy);
/* gae.(eachrow(...)) — ppo.jl:48-73,173-181 on host arrays: value/reward (nt,k) f32, terminal (nt,k) u8,
 * next_value[nt], next_done[nt]; writes adv (nt,k) and ret (nt,k) (ret may be NULL). Stateless. */
int32_t crl_gae(int32_t device, const float* value, const float* reward, const uint8_t* terminal,
                const float* next_value, const uint8_t* next_done, int32_t nt, int32_t k, float gamma, float lambda,
                int32_t mode, float* adv, float* ret);
/* The same stateless call with the kernel-flavour switches a handle carries as options (gae_seg / gae_tile / gae_nt_loads of
 * crl_ppo_set_option): gae_tile = 4 forces the streaming kernel (serial Float64 recurrence, four envs per thread), 0 lets the size
 * decide; gae_nt_loads = 2 is crl_gae's own rule (nontemporal loads from 4 M samples). ppo.jl:48-73,173-181. */
int32_t crl_gae_opt(int32_t device, const float* value, const float* reward, const uint8_t* terminal,
                    const float* next_value, const uint8_t* next_done, int32_t nt, int32_t k, float gamma, float lambda,
                    int32_t mode, float* adv, float* ret, int32_t gae_seg, int32_t gae_tile, int32_t gae_nt_loads);

/* Buffer.add!(rb, transition) — replay_buffer.jl:23-37 / ppo.jl:133-140, external-env path: slot = step (0-based). */
int32_t crl_rollout_store(crl_ppo* h, int32_t step, const float* obs, const int32_t* action, const float* logprob,
                          const float* reward, const uint8_t* terminal, const float* value);

/* reset!(env); next_obs = state(env) — ppo.jl:112-115 for the on-device vectorised env. */
int32_t crl_env_reset(crl_ppo* h);
/* The whole `for step in 1:num_steps` loop — ppo.jl:123-166 — in one launch (on-device env + policy + sampling). */
int32_t crl_rollout_run(crl_ppo* h);
int32_t crl_episode_stats_read(crl_ppo* h, crl_episode_stats* out);
/* Per-episode records of the last rollout (the fields of ppo.jl:157's "Episode Statistics"): off by default; once enabled
 * every episode end appends {return, length, global env id, step of the rollout} to a device ring of `capacity` records
 * (episodes beyond that are counted, not stored). Records come back in arrival order — sort by (step, env) for the
 * reference's logging order (ppo.jl:147-165 walks the done envs of one step in ascending order). The ring holds the LAST rollout only:
 * when a speculation guard window is replayed (crl_ppo_exact_reruns), the records are those of the replayed rollouts — the
 * speculative ones they overwrite are not kept. */
typedef struct crl_episode_record { float episode_return; int32_t episode_length; int32_t env; int32_t step; } crl_episode_record;
int32_t crl_episode_ring_enable(crl_ppo* h, int32_t capacity);
int32_t crl_episode_ring_read(crl_ppo* h, crl_episode_record* out, int32_t max_records, int32_t* n_stored, int64_t* n_episodes);
/* bootstrap + advantages + returns — ppo.jl:169-181 — on the resident buffer. */
int32_t crl_compute_gae(crl_ppo* h);
/* b_inds = shuffle(b_inds) — ppo.jl:194. epoch_id keys the counter-based stream. */
int32_t crl_shuffle(crl_ppo* h, uint64_t epoch_id);
/* per-minibatch advantage mean/std of the current permutation (ppo.jl:221), all num_minibatches at once;
 * all-reduced across ranks when a communicator is attached. Must follow crl_shuffle / a CRL_F_PERM write. */
int32_t crl_adv_stats(crl_ppo* h);
/* The two halves of crl_adv_stats for hosts that run the exchange themselves (and for the single-GPU test of the
 * data-parallel arithmetic): local sums → CRL_F_ADV_SUMS, then mean/std from whatever CRL_F_ADV_SUMS holds. */
int32_t crl_adv_stats_local(crl_ppo* h);
int32_t crl_adv_stats_finish(crl_ppo* h);
/* One `for start in 1:minibatch_size:batch_size` body — ppo.jl:197-251: loss, gradient, (all-reduce), per-array
 * ClipNorm(0.5) + Adam(eta). apply_update=0 stops after the gradient (parity tests). stats may be NULL (no sync). */
int32_t crl_ppo_update_minibatch(crl_ppo* h, int32_t mb, double eta, int32_t apply_update, crl_ppo_stats* stats);
/* n_iters passes of the ppo.jl:117-253 loop body, fully on device. stats (may be NULL) receives
 * update_epochs*num_minibatches records of the LAST iteration. */
int32_t crl_ppo_iterate(crl_ppo* h, int32_t n_iters, crl_ppo_stats* stats);
int32_t crl_ppo_iteration(const crl_ppo* h, int64_t* it);
/* The same loop body for a host that LOGS every update — ppo.jl:147-165 ("Episode Statistics") and :246-248 ("Training Statistics") — without making the GPU wait for
 * it. crl_ppo_iterate_async enqueues ONE iteration and hands back the records of the iteration BEFORE it (prev->iteration = its 0-based index; -1 on the first
 * call): at the end of every iteration one launch gathers its update_epochs * num_minibatches loss records, its episode statistics, its per-episode ring
 * (crl_episode_ring_enable; prev_ring receives up to max_ring records in arrival order), the value-loss speculation flag and the error words into a slot that
 * travels to pinned host memory on the stream; the host picks it up one call later, behind the launches of the next iteration. A host loop logs one iteration late
 * — the record stream (names, keys, order, global_step) is the reference's — and ends with crl_ppo_drain, which returns the last iteration's records (iteration =
 * -1: nothing pending). A failed speculation seen in a slot repeats the guard window exactly before the records are handed out, like every other read-back.
 * Measured against crl_ppo_iterate(h, 1, stats) + crl_episode_stats_read per update: 10.45 -> 10.21 ms per iteration at 65536 envs, 2.04 -> 1.86 at 8192, 1.65 -> 1.45 at 4096
 * (scripts/readback_cost.py). prev_stats may be NULL; the blocked shuffle limits update_epochs to 8 here. */
typedef struct crl_ppo_iteration_report {
  int64_t iteration;            /* which iteration the records belong to, -1 = none */
  crl_episode_stats episodes;   /* ppo.jl:147-162 aggregated over its rollout */
  int64_t n_episodes;           /* episodes that ended in its rollout */
  int32_t n_ring;               /* per-episode records copied to prev_ring */
  int32_t pad;
} crl_ppo_iteration_report;
int32_t crl_ppo_iterate_async(crl_ppo* h, crl_ppo_iteration_report* prev, crl_ppo_stats* prev_stats, crl_episode_record* prev_ring, int32_t max_ring);
int32_t crl_ppo_drain(crl_ppo* h, crl_ppo_iteration_report* last, crl_ppo_stats* last_stats, crl_episode_record* last_ring, int32_t max_ring);
/* how many iterations had their update phase re-run with the exact value-loss pass under data parallelism: the fused
 * kernels speculate on u = mean(v - R^2) <= 0 (ppo.jl:232-237); with an RCCL communicator a failed speculation restores the
 * parameters / Adam state of the iteration's start and repeats its optimiser steps exactly (two more small all-reduces each) */
int32_t crl_ppo_exact_reruns(const crl_ppo* h, int64_t* n);

/* Data parallelism over num_envs: one RCCL communicator per handle (one process per GPU). The 128-byte id comes
 * from crl_comm_unique_id on rank 0 and is broadcast by the host launcher. Gradients (+ advantage statistics) are
 * all-reduced once per optimiser step (ppo.jl:250 cadence) and averaged; ClipNorm/Adam then run replicated. */
int32_t crl_comm_unique_id(uint8_t id[128]);
int32_t crl_comm_init(crl_ppo* h, const uint8_t id[128], int32_t world_size, int32_t rank);
/* Which RCCL the exchange runs on: the file the loaded ncclAllReduce lives in (dladdr; NUL-terminated into path[0..path_cap)) and ncclGetVersion's
 * code (e.g. 22203). Loads librccl if it is not loaded yet; stateless. A launcher records it per rank (bench.py: comm.rccl_path / rccl_version)
 * so that a multi-GPU run shows that every rank — and a host framework that maps its own copy — uses one library. No reference counterpart. */
int32_t crl_comm_info(char* path, size_t path_cap, int32_t* version);
/* The same data-parallel exchange without RCCL: a one-shot all-reduce over peer-mapped mailboxes (csrc/peer.hip) — every rank
 * pushes its message into every peer's mailbox over xGMI and adds the world_size slots in rank order, one launch per message
 * (the 36.6 KB gradient message of a 3 ms shard iteration is latency-bound on a ring). crl_comm_peer_export allocates this
 * rank's mailbox and returns its 64-byte hipIpcMemHandle_t; the host launcher all-gathers the handles (that all-gather is the
 * barrier the protocol needs) and hands all world_size * 64 bytes, in rank order, to crl_comm_peer_attach. world_size <= 16;
 * ranks may share a GPU (that is how a 1-GPU box runs the multi-rank tests). A rank that stops answering raises a sticky
 * time-out error (option peer_timeout_ms, default 20000) reported by crl_sync / crl_ppo_iterate instead of hanging the GPU. */
int32_t crl_comm_peer_export(crl_ppo* h, int32_t world_size, int32_t rank, uint8_t handle[64]);
int32_t crl_comm_peer_attach(crl_ppo* h, const uint8_t* handles);
/* Declares this handle one of `world_size` shards WITHOUT attaching a communicator: every 1/M uses the global
 * minibatch size and the host sums the per-shard gradient messages (CRL_F_GRADS) itself. */
int32_t crl_comm_init_external(crl_ppo* h, int32_t world_size, int32_t rank);

/* Detaches whatever exchange is attached (RCCL communicator, peer mailboxes, or the external-exchange declaration) and returns the
 * handle to a single-shard configuration; a launcher uses it to fall back from a partially failed RCCL initialisation to the peer
 * all-reduce (cleanrl.jl_amd/dist.py: attach_comm). No reference counterpart (the reference is single-process). */
int32_t crl_comm_destroy(crl_ppo* h);

/* Per-handle options: every switch that selects a kernel flavour or changes numerics (earlier rounds: process-wide CRL_*
 * environment variables). Integer-valued, by name; unknown names and out-of-range values are errors. The defaults are what
 * bench.py measures. crl_ppo_option_count / crl_ppo_option_name enumerate them.
 *   gemm                    2 = 64x64 products as fp16x2 split operands (default); 1 = bf16x3 everywhere — the fallback flavour, which
 *                           a launch also takes by itself, per role, when a hidden-layer weight leaves the fp16 window (|w| >= 255)
 *   rollout_split (4)       small-shard rollout kernel, waves per 32-env tile: 4 = by size (default) — six (the actor's hidden rows over four waves on 16x16x32 products,
 *                           the critic's over two: rollout_split6_kernel) for shards up to rollout_split_max_tiles; 3 = six, 1 = three, 2 = two, 0 = one
 *   rollout_split_max_tiles largest shard, in 32-env tiles, the split kernels take (512)
 *   rollout_stagger         start delay of waves 4-7 of an 8-wave rollout block, units of 1024 clocks (6)
 *   gae_fuse                1 = inside crl_ppo_iterate the compat-mode GAE is the tail of the rollout kernel (default), 0 = own launch
 *   shuffle_overlap         1 = epoch permutations are drawn on a second stream next to the rollout (default)
 *   guard_window            iterations crl_ppo_iterate enqueues between two read-backs of the value-loss speculation flag (8)
 *   update_stagger, actor_block_pct   launch-shape knobs of the update pass (3, 53)
 *   adv_seq (1)             per-minibatch advantage sums of crl_ppo_iterate: 1 = one sequential pass over the advantages that finds a sample's minibatch through
 *                           the blocked shuffle's bucket table, with the bucket digit the shuffle stored per sample and epoch; 2 = the same with the digit
 *                           recomputed (one Philox call per sample and epoch); 0 = gathers through the finished permutations
 *   comm_force              1 = crl_comm_init with world_size 1 still creates an RCCL communicator (1-GPU test of that path)
 *   peer_timeout_ms         in-kernel time-out of the peer all-reduce (20000)
 *   wide_gemm               layer-wise path, 256-wide layers: 2 = fp16x2 (default), 1 = bf16x3, 0 = f32 MFMA
 *   wide_tanh_rational      1 = the layer-wise path evaluates NNlib's rational tanh_fast everywhere (default 0)
 *   gae_seg, gae_tile       standalone GAE kernel: steps per segment (8 / 16; 4 / 8 / 16 = window depth of the streaming kernel) / kernel shape, 0 = automatic:
 *                           gae_tile = 8 / 16 / 32 / 64 the segmented kernel with that many envs per block; 128 / 256 its two-envs-per-thread form (32 / 64 env
 *                           pairs per block, 8-byte accesses: what 4096 envs or more take by themselves when num_envs is even); 4 / 2 / 1 the streaming kernel
 *                           (that many envs per thread, serial Float64 recurrence: what batches of 33 M samples or more with >= 262144 envs take by themselves)
 *   gae_nt_loads (2)        standalone GAE kernel: 1 = nontemporal input loads (inputs not in the caches), 0 = cached, 2 = 1 for an external env with 4 M samples or more per rollout, else 0
 *   wide_rollout_persist (2)  layer-wise path, 2x256 fp16x2, obs_dim <= 16: the whole rollout as one launch — 2 = producer / consumer form (64 envs per
 *                           block, weight slabs by LDS-DMA, layer 1 on the matrix pipe; num_envs % 64 == 0, n_act <= 8, else 1), 1 = the first
 *                           one-launch kernel (32 envs per block), 0 = three launches per step
 *   update_xcd_align (1)    update kernel: tile t is worked on by blocks ≡ t (mod 8) of both roles (same XCD / L2 for a record's two readers)
 *   update_prio_small (0)   update kernel, launches with fewer than 16 tiles per wave (shards of 8192 envs and below): wave-priority rule of a
 *                           wave pair on one SIMD — 0 none (default), 1 the large-launch feedback rule, 2 static (younger wave first), 3 alternate
 *                           per tile.  Measured: 3 is -1..-2 % at 8192 envs and +1 % at 4096, the others neutral or worse
 *   wide_fuse (3)           layer-wise path, 2x256 fp16x2, obs_dim <= 16: the update pass runs the tile-resident fused kernels (csrc/wide_fused.hpp):
 *                           3 = forward, backward and a weight-gradient kernel that regenerates h1 (h1 is never stored), 2 = forward and backward,
 *                           1 = forward only, 0 = one launch per layer
 *   wide_wgrad_full (1)     the h1-regenerating weight-gradient kernel: 1 = one 256x256 tile per block, δ2 read once (default); 0 = 256x128 output tiles
 *   wide_fuse_pc (1)        the fused forward in its producer / consumer form (four MFMA-only waves, four staging waves, persistent blocks); 0 = the
 *                           symmetric first version
 *   wide_fwd_wbufs (2)      the producer / consumer forward's weight buffers in LDS: 2 = the producers fetch the next slab (default); 3 = the consumers fetch
 *                           the weight slab AFTER NEXT behind their own MFMAs (two slabs of lead for the LDS-DMA; n_act <= 6) — measured equal (DESIGN §3b:
 *                           the slab loop is bound by the SIMD's vector-issue slots, of which an LDS-DMA piece takes ~100-140 cycles whoever issues it)
 *   wide_d2_split (1)       wide_fuse = 3 with wide_wgrad_full = 1: the backward kernel hands δ2 to the weight-gradient kernel as the fp16x2 pieces it makes for
 *                           its own product (two f16 planes, one power-of-two scale per sample; the same bytes as the f32 array) and the weight-gradient kernel
 *                           multiplies straight from them (LDS-DMA + transposing LDS reads, no conversion); 0 = δ2 as f32, split again by its reader
 *   wide_rs (27)            2x256 fp16x2, register-stationary kernels (csrc/wide_rs.hpp: a network's 256 KB of W2 pieces live in the eight waves' registers
 *                           for the whole launch, nothing but the 32-sample activation tile moves): bit 0 = the update pass's forward (wide_rs_fwd_kernel,
 *                           obs_dim a multiple of 4; else the producer / consumer kernel), bit 1 = the rollout (wide_rs_rollout_kernel: the actor in the
 *                           step loop, env state in LDS; the critic as ONE batched forward over the stored observations behind it — ppo.jl:128 evaluates it
 *                           on the observation the buffer keeps), bit 2 = that critic pass on the register-stationary forward instead of the producer /
 *                           consumer kernel (measured slower), bit 3 = the update pass's backward (wide_rs_bwd_kernel: W2ᵀ stationary as the B operand, h2 /
 *                           δ3 / observations by LDS-DMA, layer 1 recomputed on the matrix pipe; with wide_d2_split = 1, obs_dim a multiple of 4), bit 4 = that kernel
 *                           also forms dW3 (h2 and δ3 are on chip: the two sweeps over h2 — 1.07 GB per optimiser step — are not launched); 0 = round 5's kernels
 *   wide_rs_actor_pct (52)  register-stationary backward: share of the CUs whose blocks take the actor's tiles (its δ2 staging costs n_act head rows against the
 *                           critic's one): 676 / 666 / 679 / 717 µs per launch at 50 / 52 / 54 / 56 % (C3)
 *   update_tile (0)         update pass of the 4 / 2 / 64 path: 32 = 32-sample tiles (update_x2_kernel), 16 = 16-sample tiles at three waves per SIMD (update16.hpp: one early-exit
 *                           repair launch redoes a minibatch as bf16x3 when a tile misses the carried weight-gradient scale or a weight leaves the fp16 window; 17 = the same with
 *                           every tile reporting a miss: test hook), 0 = by launch size — which currently means 32 everywhere: the 16-sample kernel measured slower at every
 *                           size it was built for (profiles/r06_update_tile16_ab.txt)
 *   fuse_optim (1)          speculative step of the 4 / 2 / 64 path: gradient reduction + ClipNorm + Adam as ONE launch (reduce_optim_kernel; 0 = two launches) — on one
 *                           GPU, and under data parallelism over the peer mailboxes, where the same launch also runs the exchange (reduce -> push -> wait ->
 *                           rank-order sum -> ClipNorm + Adam). Taken only where the launch's whole grid can be resident (checked at crl_ppo_create) and never
 *                           with an RCCL communicator, a host-side exchange or the inline value-loss fix-up
 * Read-only through crl_ppo_get_option: gemm_fallback_seen (1 once any launch of the fused 4/2/64 path took the bf16x3 fallback; the
 * layer-wise path needs none: it scales its fp16x2 weight pieces by the largest |w| of the layer at every optimiser step).
 * The environment variable CRL_OPTIONS="key=value,key=value" applies options at crl_ppo_create (shell-driven experiments). */
int32_t crl_ppo_set_option(crl_ppo* h, const char* key, int64_t value);
int32_t crl_ppo_get_option(crl_ppo* h, const char* key, int64_t* value);
int32_t crl_ppo_option_name(int32_t index, const char** name, int64_t* dflt);
int32_t crl_ppo_option_count(int32_t* n);   /* number of options: enumerate with crl_ppo_option_name(0 … n-1) */

/* Profiling: HIP-event timing of each kernel class on the handle's stream (bench.py roofline). on = 1: every class, events
 * recorded around the launches (extra packets between dependent kernels: a breakdown, not a throughput run); on = 2: only the
 * classes whose events ride on the dispatch itself — the update kernel of the fused path — which costs nothing, plus the gradient
 * all-reduces of a data-parallel run (recorded events: 32 packets per iteration); 0 = off. */
enum crl_kernel_id { CRL_K_ROLLOUT = 0, CRL_K_GAE = 1, CRL_K_SHUFFLE = 2, CRL_K_ADV_STATS = 3, CRL_K_UPDATE = 4,
                     CRL_K_REDUCE = 5, CRL_K_OPTIM = 6, CRL_K_ALLREDUCE = 7, CRL_K_PACK = 8, CRL_K_PERMUTE = 9,
                     CRL_K_COUNT = 10 };
/* Measurement entry (bench.py roofline_gae.beyond_cache, scripts/bench_gae.py): the standalone GAE scan (the kernel behind crl_gae /
 * crl_compute_gae, ppo.jl:48-73,173-181) on synthetic device-resident inputs of ANY size — value ~ U(-10,10), reward 0 with P 0.02,
 * terminal ~ B(0.02) — `reps` timed launches (HIP events on the dispatch), each followed by a timed launch of a hand-written float4
 * streaming copy that moves the same number of bytes (17 B per (env,step) + 5 B per env; half read, half written): the ceiling.
 * seg / tile / nt_loads as the options gae_seg / gae_tile / gae_nt_loads (0 / 0 / 0 or 1); flush_mb > 0 fills that many MiB before
 * every timed launch (sizes that fit the 256 MiB Infinity Cache); gae_ms / copy_ms receive `reps` values each. */
int32_t crl_gae_bench(int32_t device, int32_t nt, int32_t k, int32_t seg, int32_t tile, int32_t nt_loads, int32_t flush_mb,
                      int32_t reps, double* gae_ms, double* copy_ms);
/* Measurement entry (bench.py `clock`): the shader clock the GPU sustains under vector load — one 8-wave block per CU runs independent v_fma_f32 chains
 * for span_ms of the constant 100 MHz counter (s_memrealtime) and reports shader cycles (s_memtime) per microsecond: median / min / max over all waves,
 * in MHz. Boxes of one pool differ by more than a round's kernel work moves the headline; a line that carries its own clock can be normalised. */
int32_t crl_clock_probe(int32_t device, double span_ms, double* median_mhz, double* min_mhz, double* max_mhz);
/* Measurement entry (scripts/product_error.py → profiles/<tag>_product_error.json): C = A·B for the same Float32 host operands through every way the
 * hidden-layer products of networks.jl:6-13 could be multiplied on this GPU — flavour 0 = fp16x2 split operands (the default of option gemm; scale_a,
 * scale_b or a per-column col_scale[cols] are the exact powers of two the production kernels use), 1 = bf16x3 (gemm = 1), 2 = v_mfma_f32_32x32x2_f32,
 * 3 = a sequential v_fma_f32 chain per element (a scalar f32 matmul) — so that their errors against a Float64 product can be compared on real operands.
 * A [rows][K] row-major, B [cols][K]; K is cut into `chunks` pieces with one f32 accumulator each: C [chunks][cols][rows]. rows / cols multiples of 32,
 * K / chunks a multiple of 16. */
int32_t crl_product_probe(int32_t device, int32_t flavour, const float* A, const float* B, int32_t rows, int32_t cols, int32_t K, int32_t chunks,
                          float scale_a, float scale_b, const float* col_scale, float* C);
int32_t crl_prof_enable(crl_ppo* h, int32_t on);
int32_t crl_prof_read(crl_ppo* h, int32_t kernel_id, double* total_ms, int64_t* launches);
int32_t crl_prof_reset(crl_ppo* h);

/* =====================================================================================================
 * A2C: src/algorithms/a2c.jl on the GPU. One on-device CartPoleEnv{Float64}; the reference's
 * numeric regime (Float32 weights, Float64 observations ⇒ Float64 activations / losses / cotangents, a2c.jl:35,55-56)
 * is kept: these kernels are plain Float64 VALU code — a training batch is ≤ 2·min_replay_size samples of a 2x64 MLP
 * (≈0.1 GFLOP), nothing for the matrix pipes to win.
 * ===================================================================================================== */
typedef struct crl_a2c_config {   /* A2CConfig, a2c.jl:1-10 */
  double lr;                      /* a2c.jl:4 */
  int64_t total_timesteps;        /* a2c.jl:6 */
  int32_t min_replay_size;        /* a2c.jl:7; must be >= max_steps + 1 so the 2x buffer never wraps inside an update batch */
  int32_t max_steps;              /* a2c.jl:35 CartPoleEnv(max_steps=500) */
  double gamma;                   /* a2c.jl:9 */
  uint64_t seed;
} crl_a2c_config;
typedef struct crl_a2c_train_stats { double actor_loss, critic_loss; int32_t n, trained; } crl_a2c_train_stats;  /* a2c.jl:100 */
typedef struct crl_a2c_episode { double episode_return; int64_t episode_length, global_step; } crl_a2c_episode;  /* a2c.jl:106 */
typedef struct crl_a2c crl_a2c;

/* a2c.jl:32-52: networks (parameters are uploaded by crl_a2c_write_params, same flat layout as PPO: obs 4 / act 2 / 2x64),
 * Optimiser(ClipNorm(0.5), Adam(lr)) state, ReplayBuffer(capacity = 2*min_replay_size), reset!(env). */
int32_t crl_a2c_create(const crl_a2c_config* cfg, int32_t device, crl_a2c** out);
int32_t crl_a2c_destroy(crl_a2c* h);
int32_t crl_a2c_param_count(const crl_a2c* h, int64_t* n);
int32_t crl_a2c_write_params(crl_a2c* h, const float* params, size_t n);
int32_t crl_a2c_read_params(crl_a2c* h, float* params, size_t n);
/* `actor, critic = Networks.make_actor_critic(env)` — a2c.jl:37: crl_make_actor_critic(4, 2, 64, seed) uploaded. crl_a2c_run_until_update
 * on a handle whose parameters were neither written nor initialised is an error (a fresh handle holds zeros). */
int32_t crl_a2c_init_params(crl_a2c* h, uint64_t seed);
/* env state (4 doubles), global_step, current buffer size */
int32_t crl_a2c_read_env(crl_a2c* h, double* state4, int64_t* global_step, int32_t* rb_size);
/* replay buffer columns 1..size (a2c.jl:77 `x[:, 1:rb.size]`): state (4,size) Float64, action 0-based, reward, terminal */
int32_t crl_a2c_read_buffer(crl_a2c* h, double* state, int32_t* action, double* reward, uint8_t* terminal, int32_t capacity);
/* a2c.jl:53-111 `for global_step in 1:total_timesteps` body, run on the device until ONE training update has happened
 * (stats->trained = 1), max_env_steps were taken, or total_timesteps is reached. "Episode Statistics" records
 * (a2c.jl:106) of the episodes that ended are written to eps[0..*n_eps) (at most max_eps; later ones are dropped).
 * *steps_taken = env steps of this call. */
int32_t crl_a2c_run_until_update(crl_a2c* h, int64_t max_env_steps, crl_a2c_train_stats* stats, crl_a2c_episode* eps,
                                 int32_t max_eps, int32_t* n_eps, int64_t* steps_taken);
/* discounted_future_rewards(rewards, terminals, final_value, γ) (a2c.jl:13-24) on host vectors */
int32_t crl_a2c_discounted_future_rewards(int32_t device, const double* rewards, const uint8_t* terminals, int32_t n,
                                          double final_value, double gamma, double* out);

/* =====================================================================================================
 * DQN: src/algorithms/dqn.jl on the GPU. One on-device CartPoleEnv{Float64} (max_steps = 200),
 * q / target networks Chain(Dense(4,120,relu), Dense(120,84,relu), Dense(84,2)) (dqn.jl:22-26) with Float32 weights and
 * Float64 arithmetic like the reference, replay ring, ε-greedy schedule, minibatch drawn without replacement, TD target,
 * Flux.mse, Adam (no ClipNorm), hard target copy — a call enqueues the whole chunk of the `for global_step` loop (one fixed
 * ten-launch cycle per train_freq steps) and reads nothing back until it ends.
 * ===================================================================================================== */
typedef struct crl_dqn_config {   /* DQNConfig, dqn.jl:1-19 */
  int64_t log_frequency, total_timesteps, buffer_size, min_buff_size;
  double lr;
  int64_t train_freq, target_net_freq, batch_size;
  double gamma, epsilon_start, epsilon_end, epsilon_duration;
  int32_t max_steps;              /* dqn.jl:37 CartPoleEnv(): 200 */
  int32_t pad;
  uint64_t seed;
} crl_dqn_config;
typedef struct crl_dqn_episode { double episode_return; int64_t episode_length, global_step; double epsilon; } crl_dqn_episode; /* dqn.jl:88 */
typedef struct crl_dqn_loss_record { int64_t global_step; double loss; } crl_dqn_loss_record;                                  /* dqn.jl:116 */
typedef struct crl_dqn_status { double env_state[4]; int64_t global_step, rb_size, n_updates; double last_loss; } crl_dqn_status;
typedef struct crl_dqn crl_dqn;
#define CRL_DQN_PARAM_COUNT 10934 /* W1(120,4) b1 W2(84,120) b2 W3(2,84) b3 — Flux.params(q_net) order, (out,in) col-major */

int32_t crl_dqn_create(const crl_dqn_config* cfg, int32_t device, crl_dqn** out);          /* dqn.jl:34-56 */
int32_t crl_dqn_destroy(crl_dqn* h);
int32_t crl_dqn_write_params(crl_dqn* h, const float* q_params, size_t n);                  /* q_net; target_net = deepcopy(q_net) */
int32_t crl_dqn_read_params(crl_dqn* h, float* q_params, float* target_params, size_t n);   /* target_params may be NULL */
/* `q_net = make_nn(env)` — dqn.jl:22-26,39: Dense(4,120,relu) → Dense(120,84,relu) → Dense(84,2) with Flux's default glorot_uniform
 * weights and zero biases, flat in Flux.params(q_net) order; stateless and host-only. n must be CRL_DQN_PARAM_COUNT. */
int32_t crl_dqn_make_nn(uint64_t seed, float* out, size_t n);
/* dqn.jl:39-40 for a handle: crl_dqn_make_nn uploaded to q_net and target_net. crl_dqn_run / crl_dqn_q_values on a handle whose
 * parameters were neither written nor initialised are errors (a fresh handle holds zeros). */
int32_t crl_dqn_init_params(crl_dqn* h, uint64_t seed);
int32_t crl_dqn_status_read(crl_dqn* h, crl_dqn_status* out);
/* dqn.jl:57-119: up to max_env_steps iterations of the loop (or to total_timesteps), enqueued without host read-backs. Episode records
 * (dqn.jl:88) and the "Training Statistics" losses of steps that are multiples of log_frequency (dqn.jl:115-117) come
 * back in eps / losses (entries beyond max_* are dropped). */
int32_t crl_dqn_run(crl_dqn* h, int64_t max_env_steps, crl_dqn_episode* eps, int32_t max_eps, int32_t* n_eps,
                    crl_dqn_loss_record* losses, int32_t max_losses, int32_t* n_losses, int64_t* steps_taken);
/* q_net(obs) for n observations (4, n) Float64 → (2, n) Float64 (dqn.jl:64) */
int32_t crl_dqn_q_values(crl_dqn* h, const double* obs, int32_t n, double* q);

#ifdef __cplusplus
}
#endif
#endif
