// ppo_ctx.hpp — the opaque crl_ppo handle: every buffer of the path resident in HBM, one HIP stream.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/cleanrl_hip.h"

namespace crl {

void set_error(const std::string& msg);

#define CRL_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess) {                                                                                   \
      crl::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" +      \
                     std::to_string(__LINE__) + ")");                                                         \
      return 1;                                                                                               \
    }                                                                                                         \
  } while (0)

// Device-side mirror of the hyper-parameters kernels need
struct DevCfg {
  int nt, k, B, M, nmb, D, A;
  float gamma, lambda, clip, ent_coeff, v_coef;
  int clip_vloss, gae_mode, env_kind, stale_obs;
  uint32_t env_id_offset;
  uint64_t seed;
};

// One sample of the flat batch as the update kernels read it: 64 B = one HBM sector-pair, so a random gather through the
// permutation costs one fetch per sample (six separate arrays cost six). Quarter 0 = obs, 1 = what only the actor reads,
// 2 = what only the critic reads (ppo.jl:203-211 gathers the same six fields).
struct alignas(16) SampleRec {
  float x[4];
  int32_t act; float old_lp, adv, pad0;
  float old_v, ret, pad1, pad2;
  float pad3[4];
};
static_assert(sizeof(SampleRec) == 64, "SampleRec is one 64-byte record");

// Per-handle options (crl_ppo_set_option / crl_ppo_get_option, include/cleanrl_hip.h). They replace the process-wide CRL_*
// environment switches of rounds 1-2: a selector that changes numerics or picks a kernel is state of ONE handle, can be changed
// between calls, is testable in-process and is reported by bench.py. The table (name, default, range) lives in api.cpp.
enum OptId {
  OPT_GEMM = 0,               // 2 = fp16x2 split products (default), 1 = bf16x3 (the documented fallback flavour)
  OPT_ROLLOUT_SPLIT,          // small-shard rollout: 4 = by size (default: six waves per tile up to rollout_split_max_tiles), 3 = six waves, 1 = three, 2 = two, 0 = one wave per tile
  OPT_ROLLOUT_SPLIT_MAX_TILES,// largest shard (in 32-env tiles) the split kernels take
  OPT_ROLLOUT_STAGGER,        // start delay of waves 4-7 of an 8-wave rollout block (units of 1024 clocks)
  OPT_GAE_FUSE,               // 1 = compat-mode GAE rides on the rollout kernel's tail inside crl_ppo_iterate
  OPT_SHUFFLE_OVERLAP,        // 1 = epoch permutations drawn on the second stream next to the rollout
  OPT_GUARD_WINDOW,           // iterations per speculation guard window (api.cpp)
  OPT_UPDATE_STAGGER,         // start delay of waves 4-7 of an update block
  OPT_ACTOR_BLOCK_PCT,        // share of update blocks given to the actor role
  OPT_ADV_SEQ,                // advantage sums from one sequential pass over adv (blocked shuffle tables): 1 = with the bucket digits the shuffle stored (default), 2 = recomputed (one Philox call per sample and epoch), 0 = gathers through the permutations
  OPT_COMM_FORCE,             // 1 = crl_comm_init with world_size 1 still creates an RCCL communicator (1-GPU test of the path)
  OPT_PEER_TIMEOUT_MS,        // in-kernel time-out of the peer all-reduce
  OPT_WIDE_GEMM,              // layer-wise path, 256-wide layers: 2 = fp16x2 (default), 1 = bf16x3, 0 = f32 MFMA
  OPT_WIDE_TANH_RATIONAL,     // 1 = the layer-wise path evaluates tanh_fast everywhere (default 0: exp2 form outside the actor's rollout forward)
  OPT_GAE_SEG,                // standalone GAE kernel: steps per segment (0 = automatic)
  OPT_GAE_TILE,               // standalone GAE kernel: envs per block (0 = automatic)
  OPT_GAE_NT_LOADS,           // standalone GAE kernel: 1 = nontemporal input loads (inputs not in the caches), 0 = cached, 2 = external env and >= 4 M samples: 1
  OPT_WIDE_ROLLOUT_PERSIST,   // layer-wise path, 2x256 fp16x2: the whole rollout as one launch (0 = three launches per step)
  OPT_FUSE_OPTIM,             // 1 = single-GPU speculative steps run reduce + ClipNorm + Adam as one launch (reduce_optim_kernel)
  OPT_UPDATE_XCD_ALIGN,       // 1 = update kernel: a tile's actor and critic blocks on the same XCD (the record's second reader hits L2)
  OPT_UPDATE_PRIO_SMALL,      // update kernel, launches with < 16 tiles per wave: wave-priority rule (0 none, 1 feedback, 2 static for waves 4-7, 3 alternating)
  OPT_WIDE_WGRAD_FULL,        // h1-free weight gradient: 0 = 256x128 tiles, two blocks per CU (default), 1 = 256x256 tile per block (δ2 read once)
  OPT_WIDE_FUSE_PC,           // fused forward: 1 = producer / consumer waves, persistent (default), 0 = the symmetric first version
  OPT_WIDE_FUSE,              // layer-wise path, 2x256 fp16x2: 1 = tile-resident fused passes of wide_fused.hpp (default), 0 = one launch per layer
  OPT_WIDE_FWD_WBUFS,         // fused forward (producer / consumer): weight buffers in LDS — 2 = the producers fetch the next slab (default), 3 = the consumers fetch the slab after next (n_act <= 6; measured equal)
  OPT_WIDE_D2_SPLIT,          // fused backward -> 256x256 weight gradient: 1 = δ2 travels as the backward's own fp16x2 pieces (two f16 planes + a scale per sample), 0 = as f32
  OPT_UPDATE_TILE,            // update pass of the 4 / 2 / 64 path: 32 = 32-sample tiles (update_x2_kernel), 16 = 16-sample tiles at three waves per SIMD (update16.hpp), 0 = by launch size (see update.hip)
  OPT_WIDE_RS,                // 2x256 fp16x2: register-stationary kernels of wide_rs.hpp — bit 0 = the update pass's forward, bit 1 = the rollout (actor in the loop, critic as one batched pass behind it)
  OPT_WIDE_RS_ACTOR_PCT,      // register-stationary backward: share of the CUs given to the actor's blocks (its tile costs n_act head rows against one)
  OPT_COUNT
};

struct ProfSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0;
  int64_t launches = 0;
};

}  // namespace crl

struct crl_ppo {
  crl_ppo_config cfg;
  crl::DevCfg dc;
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream: crl_ppo_iterate draws the epoch permutations (which depend on nothing the rollout produces) next to the
  // rollout kernel, whose 128 dependent steps leave most issue slots and all of the memory system idle
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int64_t P = 0, Pa = 0, Pc = 0;  // total / actor / critic parameter counts
  int64_t iteration = 0;
  int64_t exact_reruns = 0;        // iterations whose update phase was re-run exactly under data parallelism (Q4)
  bool env_ready = false;          // crl_env_reset has run (crl_ppo_iterate / crl_rollout_run do it on first use)
  bool params_set = false;         // CRL_F_PARAMS was written or crl_ppo_init_params ran: ppo.jl:87 has happened (a fresh handle holds zeros, and zeros never train)
  int64_t num_updates = 1;

  // rollout buffer, Julia (·, nt, k) column-major (replay_buffer.jl:15-18)
  float* obs = nullptr; int32_t* action = nullptr; float* logprob = nullptr; float* reward = nullptr;
  uint8_t* terminal = nullptr; float* value = nullptr; float* adv = nullptr; float* ret = nullptr;
  // loop state (ppo.jl:106-115)
  float* env_state = nullptr; int32_t* env_t = nullptr; float* cur_obs = nullptr; uint8_t* next_done = nullptr;
  float* ep_return = nullptr; int32_t* ep_length = nullptr; float* next_value = nullptr;
  double* ep_stats = nullptr;  // [4] episodes, return_sum, length_sum, return_max
  crl_episode_record* ep_ring = nullptr; uint32_t* ep_ring_count = nullptr; int ep_ring_cap = 0;  // opt-in per-episode records
  // optimiser
  float* params = nullptr; float* adam_m = nullptr; float* adam_v = nullptr;  // the gradient lives in comm_buf[0..P)
  double* betap = nullptr;     // [24]
  double* optim_part = nullptr;  // per-slice Σg² of the sliced optimiser step (large networks); [12][blocks] of the fused reduce + optimiser launch
  unsigned* ticket = nullptr;    // grid meeting point of reduce_optim_kernel: [0] counts arrivals, never reset; [1] sticky time-out flag
  bool fuse_optim_fits = false;  // the whole grid of reduce_optim_kernel can be resident on this device (checked at crl_ppo_create)
  long fuse_optim_capacity = 0;  // blocks of reduce_optim_kernel the device holds at once (occupancy x CUs, no margin)
  unsigned ticket_target = 0;    // arrivals after the launch being enqueued
  // One permutation per update epoch (ppo.jl:194): crl_ppo_iterate draws all update_epochs of them right after GAE, so the
  // advantage statistics of every minibatch of the iteration are known (and all-reduced, once) before the first optimiser
  // step. `perm` / `adv_ms` point at the CURRENT slot; the host-driven entry points (crl_shuffle, …) use slot 0.
  int32_t* perm_base = nullptr;    // [update_epochs][B]
  int32_t* perm = nullptr;         // = perm_base + cur_slot * B
  int cur_slot = 0;
  // fused 4/2/64 path: the batch as 64-byte records in buffer order (records.hip); the update kernels fetch them through perm
  crl::SampleRec* recs = nullptr;      // [B]
  bool recs_dirty = true;              // a buffer field changed since the last pack
  uint32_t slot_fresh = 0;             // bit s: the advantage sums of slot s match perm[s] and recs
  double* adv_part = nullptr;          // per-block partial Σadv, Σadv² (sized in crl_ppo_create; never borrowed scratch)
  int adv_pb = 1;                      // advantage-sum blocks per minibatch
  int32_t* perm_tmp = nullptr;     // blocked Fisher–Yates: elements grouped by L1 bucket
  uint32_t* bfy_ws = nullptr;      // blocked Fisher–Yates: totals | offsets | cursors | error flag
  double* bfy_adv_part = nullptr;  // [nmb][K1][2] Σadv, Σadv² per leaf block, left behind by a fused shuffle (crl_ppo_iterate)
  int bfy_adv_parts = 0;           // K1 when the partials above belong to the current permutation, else 0
  // minibatch of every L1 bucket (low 15 bits; bit 15: the bucket straddles a minibatch boundary and its members' minibatches are in
  // bfy_mbid) — lets the advantage statistics run as ONE sequential pass over adv (records.hip) instead of a gather per epoch
  uint16_t* bfy_bucket_mb = nullptr;   // [update_epochs][BFY_MAXK1]
  uint8_t* bfy_mbid = nullptr;         // [update_epochs][B], written for members of straddling buckets only
  uint16_t* bfy_dig1 = nullptr;        // [update_epochs][B]: every sample's L1 bucket per epoch, left by bfy_l1_kernel for adv_bucket_sums_kernel (option adv_seq = 1; 2 = that pass recomputes it)
  bool bfy_dig1_valid = false;
  uint32_t bfy_tbl_slots = 0;          // bit s: the two tables of slot s describe perm[s]
  uint64_t bfy_tbl_epoch0 = 0;         // epoch id of slot 0's permutation (slot s holds epoch0 + s)
  uint32_t bfy_tbl_K1 = 0;
  bool perm_is_bijection = false;  // perm == π_key(epoch): its inverse is computable (adv-stats fast path)
  uint64_t perm_epoch = 0;
  // update workspace
  int update_blocks = 0;       // blocks per role
  float* gpart = nullptr;      // [2 roles][update_blocks][Pmax] per-block gradient partials
  double* lpart = nullptr;     // [2 roles][update_blocks][4] per-block loss partial sums
  double* adv_sums_base = nullptr;  // [update_epochs][nmb][2] Σadv, Σadv² — all-reduced under DP (one message per iteration)
  double* adv_sums = nullptr;       // current slot
  double* adv_ms_base = nullptr;    // [update_epochs][nmb][2] mean, std
  double* adv_ms = nullptr;         // current slot
  float* newv = nullptr;       // [M] critic outputs of the current minibatch (value-loss fix-up path)
  double* vfix = nullptr;      // [8] u, count(u>q), -, flag, sticky flag, fp16x2 weight-range error
  float* dscale = nullptr;     // [4] fp16x2 weight-gradient scale per role + the running largest |δ2| (mlp_x2.hpp)
  crl_ppo_stats* stats_dev = nullptr;  // [epochs*nmb]
  float* comm_buf = nullptr;   // [P + 8] gradient (+ loss scalars) message for the all-reduce
  // Data-parallel guard window (Q4): parameters / Adam state / env state at the start of a window of iterations; the sticky
  // speculation flag is read back once per window (or when the host reads results), never per iteration.
  float* snap = nullptr;       // [3P]
  double* snap_betap = nullptr;
  void* snap_env = nullptr;    // env_state | cur_obs | env_t | next_done | ep_return | ep_length | ep_stats | ring count
  size_t snap_env_bytes = 0;
  int64_t snap_iteration = 0;
  int window_count = 0;        // iterations run speculatively since the snapshot
  int window_len = 8;          // option guard_window
  // staging for host-pointer calls
  void* stage = nullptr; size_t stage_bytes = 0;
  void* pinned = nullptr; size_t pinned_bytes = 0;
  // Pipelined read-back (crl_ppo_iterate_async): what a logging host reads after every update — the loss records, the episode statistics, the per-episode ring,
  // the speculation flag and the error words — is gathered by ONE launch into a device slot at the end of an iteration, copied to pinned host memory on the
  // stream and fenced by an event; the host picks it up one iteration later, behind the next iteration's launches, so the GPU never waits for the host.
  char* status_dev[2] = {nullptr, nullptr}; char* status_host[2] = {nullptr, nullptr}; hipEvent_t status_ev[2] = {nullptr, nullptr};
  size_t status_bytes = 0; int status_ring_cap = -1;
  int64_t status_iter[2] = {-1, -1};      // iteration whose status a slot holds
  int64_t staged_last = -1, delivered_last = -1;
  bool pipelined = false;                  // the replay of a guard window re-stages the status of every iteration it repeats

  // RCCL (loaded lazily; world_size 1 = no communicator)
  void* comm = nullptr; int world = 1, rank = 0;
  bool external_comm = false;  // shards exchanged by the host (crl_comm_init_external): all-reduce calls are no-ops
  void* peer = nullptr;        // one-shot peer-mapped all-reduce (peer.hip), the alternative to the RCCL communicator
  // data-parallel optimiser step: the statistics of the all-reduced message are computed by an extra block of the optimiser launch
  bool defer_stats = false, stats_pending = false; int stats_mb = 0; crl_ppo_stats* stats_slot = nullptr;

  // generic-shape path (wide.hip): anything but obs 4 / act 2 / hidden 64, or CRL_FORCE_WIDE=1
  bool wide = false;
  void* wide_ws = nullptr;

  int64_t opt[crl::OPT_COUNT] = {};   // crl_ppo_set_option

  int prof = 0;   // 0 off, 1 every kernel class (events recorded around the launches), 2 only the kernels whose events ride ON the dispatch
  crl::ProfSlot prof_slots[CRL_K_COUNT];
};

namespace crl {
// Key of the minibatch shuffle (ppo.jl:194). Under data parallelism every shard draws its OWN permutation: the key folds in
// the shard's first global env id, so sample positions are not correlated across ranks (offset 0 = the plain seed).
inline uint64_t shuffle_seed(const crl_ppo* h) {
  return h->cfg.seed + 0x9E3779B97F4A7C15ull * (uint64_t)(uint32_t)h->cfg.env_id_offset;
}
void select_slot(crl_ppo* h, int slot);
int ensure_records(crl_ppo* h);
int ensure_stage(crl_ppo* h, size_t bytes);
int reset_dw_scale(crl_ppo* h);
// option "gemm": 2 (default) = fp16x2 products in the update kernel and the rollout's critic, bf16x3 for the rollout's actor
// (mlp_x2.hpp); 1 = bf16x3 everywhere (mlp_x3.hpp) — the fallback flavour, also taken per block when a hidden-layer weight
// leaves the fp16x2 window
inline bool gemm_x2(const crl_ppo* h) { return h->opt[OPT_GEMM] == 2; }
inline int64_t opt(const crl_ppo* h, int id) { return h->opt[id]; }

// HIP-event timing of one kernel class. attach=true: the events are handed to hipExtLaunchKernelGGL, which stamps the
// kernel's own begin/end (what rocprofv3 reports); otherwise they are recorded on the stream around the launch(es).
struct ProfScope {
  crl_ppo* h; int id; bool attach; hipEvent_t a = nullptr, b = nullptr;
  bool on;
  ProfScope(crl_ppo* h_, int id_, bool attach_ = false) : h(h_), id(id_), attach(attach_) {
    // level 2 times only the attached kind: recorded events are extra packets between dependent kernels (≈0.3 ms per iteration
    // at the headline size), which a throughput measurement should not carry
    on = h->prof == 1 || (h->prof == 2 && (attach || id == CRL_K_ALLREDUCE));   // + the 16 all-reduces of a data-parallel iteration
    if (on) {
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      if (!attach) (void)hipEventRecord(a, h->stream);
    }
  }
  ~ProfScope() {
    if (on) {
      if (!attach) (void)hipEventRecord(b, h->stream);
      h->prof_slots[id].pending.emplace_back(a, b);
    }
  }
};

// kernel launchers (each in its own translation unit)
int launch_gae(hipStream_t st, const float* value, const float* reward, const uint8_t* terminal,
               const float* next_value, const uint8_t* next_done, int nt, int k, float gamma, float lambda, int mode,
               float* adv, float* ret, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int seg = 0, int tile = 0, int nt_loads = 0);
int launch_policy_act(crl_ppo* h, const float* obs_d, const double* u_d, int n, int32_t* action_d, float* logprob_d,
                      float* value_d);
int launch_logprob_actions(crl_ppo* h, const float* obs_d, const int32_t* act_d, int n, float* logprob_d, float* ent_d);
int launch_env_reset(crl_ppo* h);
int launch_rollout(crl_ppo* h, bool fuse_gae = false);
bool rollout_can_fuse_gae(const crl_ppo* h);
int launch_next_value(crl_ppo* h);
int launch_shuffle(crl_ppo* h, uint64_t epoch_id, bool with_adv_sums = false);
int launch_shuffle_epochs(crl_ppo* h, uint64_t epoch0, int nslots);
int launch_adv_stats_sums(crl_ppo* h);
int launch_adv_bucket_sums(crl_ppo* h, int slot0, int nslots, double* part, int nblk);   // shuffle.hip; 1 = not applicable
int launch_adv_stats_finish(crl_ppo* h, int slot0 = -1, int nslots = 1);
int launch_pack_records(crl_ppo* h);
int launch_guard_copy(crl_ppo* h, const void* const* src, void* const* dst, const size_t* bytes, int n);   // records.hip: the guard snapshot as one launch
int launch_slot_adv_sums(crl_ppo* h, int slot0, int nslots);   // records.hip: Σadv, Σadv² of every minibatch of the slots
int launch_update(crl_ppo* h, int mb, crl_ppo_stats* stats_slot, bool inline_fix = true, bool with_optim = false, double eta = 0.0);
int launch_update_exact_dp(crl_ppo* h, int mb, crl_ppo_stats* stats_slot);
int launch_optim(crl_ppo* h, double eta);
int comm_allreduce(crl_ppo* h, void* buf, size_t count, bool is_double);
// wide.hip — layer-wise path for other network shapes
bool wide_shape_ok(const crl_ppo_config* cfg, std::string* why);
int wide_create(crl_ppo* h);
void wide_destroy(crl_ppo* h);
void wide_mark_params_changed(crl_ppo* h);
bool wide_x2_active(const crl_ppo* h);
int wide_policy_act(crl_ppo* h, const float* obs_d, const double* u_d, int n, int32_t* action_d, float* logprob_d, float* value_d);
int wide_logprob_actions(crl_ppo* h, const float* obs_d, const int32_t* act_d, int n, float* logprob_d, float* ent_d);
int wide_next_value(crl_ppo* h);
int wide_env_reset(crl_ppo* h);
int wide_rollout(crl_ppo* h);
int wide_update(crl_ppo* h, int mb, crl_ppo_stats* stats_slot);
void comm_destroy(crl_ppo* h);
// peer.hip — one-shot all-reduce over hipIpc-shared mailboxes
int peer_export(crl_ppo* h, int world, int rank, uint8_t handle[64]);
int peer_attach(crl_ppo* h, const uint8_t* handles);
bool peer_active(const crl_ppo* h);
int peer_ranks_on_my_device(const crl_ppo* h);   // > 1: ranks share this GPU (functional runs on a 1-GPU box)
int peer_allreduce(crl_ppo* h, void* buf, size_t count, bool is_double);
// what a kernel needs to take part in one peer exchange (peer.hip): mailbox bases, this message's sequence number, the layout
constexpr int PEER_MAX = 16;
struct PeerArgs {
  char* box[PEER_MAX];
  int world, rank;
  uint32_t seq;
  int nblk;             // flag words per (parity, rank)
  size_t slot_bytes;
  size_t data_off;      // byte offset of the slots inside a mailbox
  uint32_t* err;
  long long timeout_ticks;
};
// the next message's arguments for a kernel that runs the exchange itself (update.hip: reduce_optim_kernel<true>, 64-float chunks,
// one flag per chunk): bumps the sequence number like peer_allreduce does; `chunks` flag words are needed
int peer_next_args(crl_ppo* h, PeerArgs* out, int chunks, size_t floats);
int peer_check(crl_ppo* h);
const uint32_t* peer_err_word(const crl_ppo* h);   // device address of the exchange's sticky time-out word (nullptr: no peer exchange)
int fused_optim_fits(crl_ppo* h, bool* fits);   // update.hip: occupancy of reduce_optim_kernel's grid on this device
int fused_optim_check(crl_ppo* h);              // update.hip: sticky time-out word of its meeting point
void peer_destroy(crl_ppo* h);
inline bool has_comm(const crl_ppo* h) { return h->comm != nullptr || h->peer != nullptr; }
}  // namespace crl
