// api.cpp — the extern "C" boundary of libcleanrl_hip.so (include/cleanrl_hip.h). Host orchestration only; every
// numeric step is a HIP kernel in gae.hip / policy.hip / update.hip / optim.hip / shuffle.hip. There is no CPU fallback:
// without a GPU every entry point that computes returns an error.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "ppo_ctx.hpp"

namespace crl {
thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int comm_unique_id(uint8_t id[128]);
int comm_init(crl_ppo* h, const uint8_t id[128], int world, int rank);
int comm_info(char* path, size_t path_cap, int* version);
int launch_iota(crl_ppo* h);

// Option table of crl_ppo_set_option / crl_ppo_get_option (ids: ppo_ctx.hpp). Every switch that selects a kernel flavour or changes
// numerics is state of one handle; the process environment is read in exactly one place (CRL_OPTIONS, crl_ppo_create).
struct OptDesc { const char* name; int64_t dflt, lo, hi; };
static const OptDesc kOpts[OPT_COUNT] = {
    {"gemm", 2, 1, 2},
    {"rollout_split", 4, 0, 4},
    {"rollout_split_max_tiles", 512, 0, 1 << 20},
    {"rollout_stagger", 6, 0, 64},
    {"gae_fuse", 1, 0, 1},
    {"shuffle_overlap", 1, 0, 1},
    {"guard_window", 8, 1, 1 << 20},
    {"update_stagger", 3, 0, 64},
    {"actor_block_pct", 53, 1, 99},
    {"adv_seq", 1, 0, 2},
    {"comm_force", 0, 0, 1},
    {"peer_timeout_ms", 20000, 1, 3600000},
    {"wide_gemm", 2, 0, 2},
    {"wide_tanh_rational", 0, 0, 1},
    {"gae_seg", 0, 0, 16},
    {"gae_tile", 0, 0, 256},
    {"gae_nt_loads", 2, 0, 2},
    {"wide_rollout_persist", 2, 0, 2},
    {"fuse_optim", 1, 0, 1},
    {"update_xcd_align", 1, 0, 1},
    {"update_prio_small", 0, 0, 3},
    {"wide_wgrad_full", 1, 0, 1},
    {"wide_fuse_pc", 1, 0, 1},
    {"wide_fuse", 3, 0, 3},
    {"wide_fwd_wbufs", 2, 0, 3},
    {"wide_d2_split", 1, 0, 1},
    {"update_tile", 0, 0, 32},
    {"wide_rs", 27, 0, 31},
    {"wide_rs_actor_pct", 52, 10, 90},
};
static bool gae_seg_ok(int64_t v) { return v == 0 || v == 4 || v == 8 || v == 16; }
static bool gae_tile_ok(int64_t v) { return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64 || v == 128 || v == 256; }
static int opt_find(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < OPT_COUNT; ++i) if (std::strcmp(kOpts[i].name, key) == 0) return i;
  return -1;
}
static int opt_set(crl_ppo* h, const char* key, int64_t value) {
  const int id = opt_find(key);
  if (id < 0) { set_error(std::string("crl_ppo_set_option: unknown option '") + (key ? key : "(null)") + "'"); return 1; }
  if (value < kOpts[id].lo || value > kOpts[id].hi) {
    set_error(std::string("crl_ppo_set_option: ") + key + " = " + std::to_string(value) + " is outside [" + std::to_string(kOpts[id].lo) + ", " +
              std::to_string(kOpts[id].hi) + "]");
    return 1;
  }
  if (id == OPT_GAE_SEG && !gae_seg_ok(value)) { set_error("crl_ppo_set_option: gae_seg is 0 (automatic), 8 or 16 (4: streaming kernel only)"); return 1; }
  if (id == OPT_WIDE_FWD_WBUFS && value == 1) { set_error("crl_ppo_set_option: wide_fwd_wbufs is 0 (weight fragments to registers), 2 or 3 (LDS buffers)"); return 1; }
  if (id == OPT_UPDATE_TILE && value != 0 && value != 16 && value != 17 && value != 32) { set_error("crl_ppo_set_option: update_tile is 0 (by launch size), 16 (16-sample tiles, three waves per SIMD; 17 = the same with every tile reporting a scale miss: test hook) or 32"); return 1; }
  if (id == OPT_GAE_TILE && !gae_tile_ok(value)) { set_error("crl_ppo_set_option: gae_tile is 0 (automatic), 1 / 2 / 4 (streaming kernel, envs per thread), 8, 16, 32 or 64 (segmented kernel, envs per block), 128 or 256 (two envs per thread, 32 / 64 pairs per block)"); return 1; }
  h->opt[id] = value;
  if (id == OPT_GUARD_WINDOW) h->window_len = (int)value;
  if (id == OPT_WIDE_GEMM) wide_mark_params_changed(h);   // the packed weight copies depend on the flavour
  return 0;
}
// CRL_OPTIONS="key=value,key=value": the one environment hook left, for shell-driven experiments (scripts/, bench.py --opt goes
// through crl_ppo_set_option instead). Unknown keys and out-of-range values fail crl_ppo_create loudly.
static int opt_apply_env(crl_ppo* h) {
  const char* e = std::getenv("CRL_OPTIONS");
  if (!e || !*e) return 0;
  std::string s(e);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) { set_error("CRL_OPTIONS: expected key=value, got '" + item + "'"); return 1; }
    char* endp = nullptr;
    const long long v = std::strtoll(item.c_str() + eq + 1, &endp, 10);
    if (!endp || endp == item.c_str() + eq + 1 || *endp) { set_error("CRL_OPTIONS: value of '" + item + "' is not an integer"); return 1; }
    if (opt_set(h, item.substr(0, eq).c_str(), (int64_t)v)) return 1;
  }
  return 0;
}

// first guess of the fp16x2 weight-gradient scale (mlp_x2.hpp) before any launch has measured |δ2|: δ2 ∝ 1/M, times the
// typical head weights (actor gain 0.01, critic gain 1) and cotangents; powers of 2^8 like every later value
int reset_dw_scale(crl_ppo* h) {
  const double lm = std::log2((double)h->dc.M * (double)h->world);
  const int ka = 8 * (int)std::lround((lm + 19.0) / 8.0), kc = 8 * (int)std::lround((lm + 3.0) / 8.0);
  const float init[8] = {std::ldexp(1.0f, ka), std::ldexp(1.0f, kc), 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // G x2 | max bits x2 | miss flag of the 16-sample-tile kernel
  CRL_HIP_CHECK(hipMemcpy(h->dscale, init, sizeof(init), hipMemcpyHostToDevice));
  return 0;
}

int ensure_stage(crl_ppo* h, size_t bytes) {
  if (h->stage_bytes >= bytes) return 0;
  if (h->stage) CRL_HIP_CHECK(hipFree(h->stage));
  h->stage = nullptr; h->stage_bytes = 0;
  size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
  CRL_HIP_CHECK(hipMalloc(&h->stage, want));
  h->stage_bytes = want;
  return 0;
}

template <typename T>
static int dalloc(T** p, size_t n, bool zero = true) {
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  if (zero) CRL_HIP_CHECK(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

struct FieldRef { void* ptr; size_t bytes; };
static bool field_ref(crl_ppo* h, int f, FieldRef* out) {
  const size_t B = (size_t)h->dc.B, nt = (size_t)h->dc.nt, d = (size_t)h->dc.D, P = (size_t)h->P;
  switch (f) {
    case CRL_F_OBS: *out = {h->obs, B * d * 4}; return true;
    case CRL_F_ACTION: *out = {h->action, B * 4}; return true;
    case CRL_F_LOGPROB: *out = {h->logprob, B * 4}; return true;
    case CRL_F_REWARD: *out = {h->reward, B * 4}; return true;
    case CRL_F_TERMINAL: *out = {h->terminal, B}; return true;
    case CRL_F_VALUE: *out = {h->value, B * 4}; return true;
    case CRL_F_ADVANTAGE: *out = {h->adv, B * 4}; return true;
    case CRL_F_RETURN: *out = {h->ret, B * 4}; return true;
    case CRL_F_PERM: *out = {h->perm, B * 4}; return true;
    case CRL_F_PARAMS: *out = {h->params, P * 4}; return true;
    case CRL_F_GRADS: *out = {h->comm_buf, P * 4}; return true;
    case CRL_F_ADAM_M: *out = {h->adam_m, P * 4}; return true;
    case CRL_F_ADAM_V: *out = {h->adam_v, P * 4}; return true;
    case CRL_F_ENV_STATE: *out = {h->env_state, nt * d * 4}; return true;
    case CRL_F_CUR_OBS: *out = {h->cur_obs, nt * d * 4}; return true;
    case CRL_F_NEXT_DONE: *out = {h->next_done, nt}; return true;
    case CRL_F_ENV_T: *out = {h->env_t, nt * 4}; return true;
    case CRL_F_BETAP: *out = {h->betap, 24 * 8}; return true;
    case CRL_F_ADV_SUMS: *out = {h->adv_sums, (size_t)h->dc.nmb * 2 * 8}; return true;  // current slot
    default: return false;
  }
}

static int prof_collect(crl_ppo* h) {
  for (int k = 0; k < CRL_K_COUNT; ++k) {
    auto& s = h->prof_slots[k];
    for (auto& pr : s.pending) {
      CRL_HIP_CHECK(hipEventSynchronize(pr.second));
      float ms = 0.f;
      CRL_HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
      s.total_ms += ms; s.launches += 1;
      (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    s.pending.clear();
  }
  return 0;
}

// `perm`, `adv_sums`, `adv_ms` follow the current slot (one slot per update epoch)
void select_slot(crl_ppo* h, int slot) {
  h->cur_slot = slot;
  h->perm = h->perm_base + (size_t)slot * h->dc.B;
  h->adv_sums = h->adv_sums_base + (size_t)slot * h->dc.nmb * 2;
  h->adv_ms = h->adv_ms_base + (size_t)slot * h->dc.nmb * 2;
}

// fused path: the records are packed and the current slot's advantage sums are current
int ensure_records(crl_ppo* h) {
  if (h->wide) return 0;
  if (!h->recs_dirty && (h->slot_fresh >> h->cur_slot & 1u)) return 0;
  return launch_slot_adv_sums(h, h->cur_slot, 1);
}
}  // namespace crl

namespace crl {
// compat-mode GAE rides on the tail of the rollout kernels of the fused 4/2/64 path (policy.hip); fixed mode needs the
// bootstrap critic pass and keeps the separate launches
bool rollout_can_fuse_gae(const crl_ppo* h) {
  return opt(h, OPT_GAE_FUSE) != 0 && !h->wide && h->cfg.gae_mode == CRL_GAE_COMPAT && h->cfg.env_kind == CRL_ENV_CARTPOLE;
}
}  // namespace crl

using namespace crl;

#define CRL_GUARD(h)                                         \
  if (!(h)) { set_error("null crl_ppo handle"); return 1; } \
  CRL_HIP_CHECK(hipSetDevice((h)->device));

static int settle(crl_ppo* h);
static int check_bfy(crl_ppo* h, bool host_syncs = true);
// Every entry point that reads or mutates handle state first closes an open speculation guard window (see settle below): a
// host-driven step, an env reset or a field write issued inside a window would otherwise be undone by a later restore + replay.
// crl_ppo_iterate itself and the pure getters are the only entry points that do not.
#define CRL_GUARD_SETTLED(h) \
  CRL_GUARD(h);              \
  if (settle(h)) return 1;
// ppo.jl:87 builds the networks before anything uses them. A fresh handle holds all-zero parameters, with which h1 = h2 = 0 and every
// gradient except the head biases' is 0 for ever: computing with them is refused, never done silently.
#define CRL_NEED_PARAMS(h, who)                                                                                                        \
  if (!(h)->params_set) {                                                                                                              \
    set_error(std::string(who) + ": parameters not set — upload Flux.params(actor, critic) with crl_ppo_write(CRL_F_PARAMS) or call " \
              "crl_ppo_init_params first (ppo.jl:87; a fresh handle holds zeros)");                                                    \
    return 1;                                                                                                                          \
  }

extern "C" {

int32_t crl_version(void) { return CRL_VERSION; }
const char* crl_last_error(void) { return g_err.c_str(); }

int32_t crl_device_count(int32_t* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); return 1; }
  *n = c;
  return 0;
}

int32_t crl_ppo_create(const crl_ppo_config* cfg, int32_t device, crl_ppo** out) {
  if (!cfg || !out) { set_error("crl_ppo_create: null argument"); return 1; }
  *out = nullptr;
  if (cfg->num_envs <= 0 || cfg->num_steps <= 0 || cfg->num_minibatches <= 0 || cfg->update_epochs <= 0) {
    set_error("crl_ppo_create: num_envs, num_steps, num_minibatches, update_epochs must be positive"); return 1;
  }
  if (!cfg->normalize_advantages) {
    set_error("normalize_advantages=false is not a valid configuration (the reference errors too: ppo.jl:219-222)"); return 1;
  }
  // obs 4 / act 2 / 2x64 (the reference's CartPole shape) runs the fused register-resident kernels; every other
  // supported shape runs the layer-wise path of wide.hip (CRL_FORCE_WIDE=1 sends the CartPole shape there too)
  const bool narrow_shape = cfg->obs_dim == 4 && cfg->n_act == 2 && cfg->hidden == 64;
  const char* fw = getenv("CRL_FORCE_WIDE");
  const bool wide = !narrow_shape || (fw && atoi(fw) != 0);
  if (wide) {
    std::string why;
    if (!wide_shape_ok(cfg, &why)) { set_error("unsupported network shape: " + why); return 1; }
  }
  if (cfg->env_kind == CRL_ENV_CARTPOLE && (cfg->obs_dim != 4 || cfg->n_act != 2)) {
    set_error("env_kind = CRL_ENV_CARTPOLE needs obs_dim=4, n_act=2 (use CRL_ENV_SYNTHETIC or CRL_ENV_EXTERNAL)"); return 1;
  }
  if (cfg->env_kind != CRL_ENV_CARTPOLE && cfg->env_kind != CRL_ENV_SYNTHETIC && cfg->env_kind != CRL_ENV_EXTERNAL) {
    set_error("unknown env_kind"); return 1;
  }
  if (cfg->env_kind == CRL_ENV_SYNTHETIC && !wide) {
    set_error("env_kind = CRL_ENV_SYNTHETIC runs on the generic-shape path only (set CRL_FORCE_WIDE=1 for the 4/2/64 shape)"); return 1;
  }
  const int64_t B64 = (int64_t)cfg->num_envs * cfg->num_steps;
  if (B64 > (1ll << 30)) { set_error("batch too large"); return 1; }
  if (B64 % cfg->num_minibatches != 0) {
    set_error("num_envs*num_steps must be divisible by num_minibatches"); return 1;
  }
  if (cfg->num_steps > 1024) { set_error("num_steps > 1024 is not supported"); return 1; }
  if (cfg->num_minibatches > 1024) { set_error("num_minibatches > 1024 is not supported"); return 1; }
  if (cfg->update_epochs > 32) { set_error("update_epochs > 32 is not supported"); return 1; }
  int ndev = 0;
  CRL_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { set_error("crl_ppo_create: no such HIP device (no GPU → no CPU fallback)"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  crl_ppo* h = new (std::nothrow) crl_ppo();
  if (!h) { set_error("out of host memory"); return 1; }
  h->cfg = *cfg; h->device = device; h->wide = wide;
  for (int i = 0; i < OPT_COUNT; ++i) h->opt[i] = kOpts[i].dflt;
  h->window_len = (int)kOpts[OPT_GUARD_WINDOW].dflt;
  if (opt_apply_env(h)) { delete h; return 1; }
  DevCfg& c = h->dc;
  c.nt = cfg->num_envs; c.k = cfg->num_steps; c.B = (int)B64; c.nmb = cfg->num_minibatches; c.M = c.B / c.nmb;
  c.D = cfg->obs_dim; c.A = cfg->n_act; c.gamma = cfg->gamma; c.lambda = cfg->gae_lambda; c.clip = cfg->clip_coef;
  c.ent_coeff = cfg->ent_coeff; c.v_coef = cfg->v_coef; c.clip_vloss = cfg->clip_value_loss; c.gae_mode = cfg->gae_mode;
  c.env_kind = cfg->env_kind; c.stale_obs = cfg->stale_obs; c.env_id_offset = (uint32_t)cfg->env_id_offset; c.seed = cfg->seed;
  const int hN = cfg->hidden, d = cfg->obs_dim, A = cfg->n_act;
  h->Pa = (int64_t)hN * d + hN + hN * hN + hN + A * hN + A;
  h->Pc = (int64_t)hN * d + hN + hN * hN + hN + hN + 1;
  h->P = h->Pa + h->Pc;
  h->num_updates = cfg->total_timesteps / B64;
  if (h->num_updates < 1) h->num_updates = 1;
  int rc = 0;
  hipError_t se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { set_error(std::string("hipStreamCreate: ") + hipGetErrorString(se)); delete h; return 1; }
  if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_error("hipStreamCreate / hipEventCreate failed"); crl_ppo_destroy(h); return 1;
  }
  const size_t B = (size_t)c.B, nt = (size_t)c.nt;
  rc |= dalloc(&h->obs, B * d); rc |= dalloc(&h->action, B); rc |= dalloc(&h->logprob, B); rc |= dalloc(&h->reward, B);
  rc |= dalloc(&h->terminal, B); rc |= dalloc(&h->value, B); rc |= dalloc(&h->adv, B); rc |= dalloc(&h->ret, B);
  rc |= dalloc(&h->env_state, nt * d); rc |= dalloc(&h->env_t, nt); rc |= dalloc(&h->cur_obs, nt * d);
  rc |= dalloc(&h->next_done, nt); rc |= dalloc(&h->ep_return, nt); rc |= dalloc(&h->ep_length, nt);
  rc |= dalloc(&h->next_value, nt); rc |= dalloc(&h->ep_stats, 4);
  rc |= dalloc(&h->params, h->P); rc |= dalloc(&h->adam_m, h->P); rc |= dalloc(&h->adam_v, h->P);
  const size_t E = (size_t)cfg->update_epochs;
  rc |= dalloc(&h->betap, 24); rc |= dalloc(&h->perm_base, E * B); rc |= dalloc(&h->optim_part, (size_t)h->P / 4096 + 16 + 12 * ((size_t)h->P / 64 + 1)); rc |= dalloc(&h->ticket, 2);
  if (!wide) rc |= dalloc(&h->recs, B);   // the update kernels fetch records through the epoch's permutation: no permuted copies
  {
    // advantage-sum pass: blocks per minibatch (≈1 K samples each, at most 512); the partial-sum scratch also serves the
    // stand-alone statistics kernels (up to 512 blocks per minibatch of ONE slot)
    int pb = c.M / 1024; if (pb < 1) pb = 1; if (pb > 512) pb = 512;
    h->adv_pb = pb;
    const size_t need = E * (size_t)c.nmb * pb, alone = (size_t)c.nmb * 512;
    rc |= dalloc(&h->adv_part, 2 * (need > alone ? need : alone));
  }
  if (cfg->shuffle_mode == CRL_SHUFFLE_BLOCKED_FY) {
    size_t k1 = 1; while (k1 * 4096 < B) k1 *= 2;
    rc |= dalloc(&h->perm_tmp, E * k1 * 5632);   // padded L1 buckets (K1 x BFY_CAP), one slice per epoch slot
    rc |= dalloc(&h->bfy_ws, E * ((size_t)4 * 16384 + 8));
    rc |= dalloc(&h->bfy_adv_part, (size_t)c.nmb * k1 * 2);
    if (!wide) { rc |= dalloc(&h->bfy_bucket_mb, E * (size_t)16384); rc |= dalloc(&h->bfy_mbid, E * B); rc |= dalloc(&h->bfy_dig1, E * B); }
  }
  // update grid: two 256-thread blocks per CU, alternating roles; never more waves than tiles
  hipDeviceProp_t prop;
  CRL_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  int ntiles = (c.M + 31) / 32;
  int ub = cus;  // blocks per role
  if (ub * 4 > ntiles) ub = (ntiles + 3) / 4;
  if (ub < 1) ub = 1;
  h->update_blocks = ub;
  if (!wide) { rc |= dalloc(&h->gpart, (size_t)4 * ub * h->Pa + 4096); rc |= dalloc(&h->lpart, (size_t)4 * ub * 2); }
  rc |= dalloc(&h->adv_sums_base, E * c.nmb * 2); rc |= dalloc(&h->adv_ms_base, E * c.nmb * 2);
  rc |= dalloc(&h->newv, (size_t)c.M + 64); rc |= dalloc(&h->vfix, 8); rc |= dalloc(&h->dscale, 8);
  rc |= dalloc(&h->stats_dev, (size_t)cfg->update_epochs * c.nmb);
  rc |= dalloc(&h->comm_buf, (size_t)h->P + 8);
  rc |= dalloc(&h->snap, (size_t)3 * h->P); rc |= dalloc(&h->snap_betap, 24);
  h->snap_env_bytes = nt * d * 8 + nt * 4 * 3 + ((nt + 15) & ~(size_t)15) + 96;
  { char* p = nullptr; rc |= dalloc(&p, h->snap_env_bytes); h->snap_env = p; }
  if (rc) { crl_ppo_destroy(h); return 1; }
  if (reset_dw_scale(h)) { crl_ppo_destroy(h); return 1; }
  if (!wide && h->P <= 32768 && fused_optim_fits(h, &h->fuse_optim_fits)) { crl_ppo_destroy(h); return 1; }
  select_slot(h, 0);
  if (wide && wide_create(h)) { crl_ppo_destroy(h); return 1; }
  double bp[24];
  for (int i = 0; i < 12; ++i) { bp[2 * i] = 0.9; bp[2 * i + 1] = 0.999; }
  CRL_HIP_CHECK(hipMemcpy(h->betap, bp, sizeof(bp), hipMemcpyHostToDevice));
  if (launch_iota(h)) { crl_ppo_destroy(h); return 1; }
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  *out = h;
  return 0;
}

int32_t crl_ppo_destroy(crl_ppo* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  peer_destroy(h);
  comm_destroy(h);
  wide_destroy(h);
  void* ptrs[] = {h->obs, h->action, h->logprob, h->reward, h->terminal, h->value, h->adv, h->ret, h->env_state, h->env_t,
                  h->cur_obs, h->next_done, h->ep_return, h->ep_length, h->next_value, h->ep_stats, h->ep_ring, h->ep_ring_count, h->params,
                  h->adam_m, h->adam_v, h->betap, h->optim_part, h->ticket, h->perm_base, h->recs, h->adv_part, h->perm_tmp, h->bfy_ws, h->bfy_adv_part, h->bfy_bucket_mb, h->bfy_mbid, h->bfy_dig1, h->gpart, h->lpart,
                  h->adv_sums_base, h->adv_ms_base, h->newv, h->vfix, h->dscale, h->stats_dev, h->comm_buf, h->snap, h->snap_betap, h->snap_env, h->stage};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (int s = 0; s < 2; ++s) {
    if (h->status_dev[s]) (void)hipFree(h->status_dev[s]);
    if (h->status_host[s]) (void)hipHostFree(h->status_host[s]);
    if (h->status_ev[s]) (void)hipEventDestroy(h->status_ev[s]);
  }
  for (int k = 0; k < CRL_K_COUNT; ++k)
    for (auto& pr : h->prof_slots[k].pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int32_t crl_ppo_param_count(const crl_ppo* h, int64_t* n) {
  if (!h || !n) { set_error("null argument"); return 1; }
  *n = h->P;
  return 0;
}

int32_t crl_sync(crl_ppo* h) {
  CRL_GUARD(h);
  if (settle(h)) return 1;
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return check_bfy(h);
}


int32_t crl_ppo_write(crl_ppo* h, int32_t field, const void* host, size_t nbytes) {
  CRL_GUARD(h);
  if (settle(h)) return 1;
  FieldRef fr;
  if (!field_ref(h, field, &fr)) { set_error("crl_ppo_write: unknown field"); return 1; }
  if (nbytes != fr.bytes) { set_error("crl_ppo_write: size mismatch for field " + std::to_string(field) + ": got " +
                                      std::to_string(nbytes) + ", want " + std::to_string(fr.bytes)); return 1; }
  if (field == CRL_F_PERM) {
    // b_inds index the batch inside the kernels (records are fetched through them): an entry outside [0, B) would be an
    // out-of-bounds device read, so it is rejected here
    const int32_t* pv = static_cast<const int32_t*>(host);
    const int32_t B = h->dc.B;
    for (int32_t i = 0; i < B; ++i)
      if (pv[i] < 0 || pv[i] >= B) {
        set_error("crl_ppo_write(CRL_F_PERM): entry " + std::to_string(i) + " = " + std::to_string(pv[i]) + " is outside [0, " + std::to_string(B) + ")");
        return 1;
      }
  }
  CRL_HIP_CHECK(hipMemcpyAsync(fr.ptr, host, nbytes, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (field == CRL_F_PARAMS) { wide_mark_params_changed(h); h->params_set = true; }
  if (field == CRL_F_PERM) { h->perm_is_bijection = false; h->slot_fresh &= ~(1u << h->cur_slot); h->bfy_tbl_slots &= ~(1u << h->cur_slot); }
  if (field == CRL_F_PERM || field == CRL_F_ADVANTAGE) h->bfy_adv_parts = 0;  // a caller-supplied permutation has no closed-form inverse
  if (field == CRL_F_OBS || field == CRL_F_ACTION || field == CRL_F_LOGPROB || field == CRL_F_VALUE || field == CRL_F_ADVANTAGE ||
      field == CRL_F_RETURN) h->recs_dirty = true;
  if (field == CRL_F_ENV_STATE || field == CRL_F_CUR_OBS) h->env_ready = true;  // caller-supplied env state
  return 0;
}

int32_t crl_ppo_init_params(crl_ppo* h, uint64_t seed) {
  CRL_GUARD_SETTLED(h);
  std::vector<float> w((size_t)h->P);
  if (crl_make_actor_critic(h->cfg.obs_dim, h->cfg.n_act, h->cfg.hidden, seed, w.data(), w.size())) return 1;
  return crl_ppo_write(h, CRL_F_PARAMS, w.data(), w.size() * sizeof(float));
}

int32_t crl_ppo_read(crl_ppo* h, int32_t field, void* host, size_t nbytes) {
  CRL_GUARD(h);
  if (settle(h)) return 1;
  FieldRef fr;
  if (!field_ref(h, field, &fr)) { set_error("crl_ppo_read: unknown field"); return 1; }
  if (nbytes != fr.bytes) { set_error("crl_ppo_read: size mismatch for field " + std::to_string(field) + ": got " +
                                      std::to_string(nbytes) + ", want " + std::to_string(fr.bytes)); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(host, fr.ptr, nbytes, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int32_t crl_policy_act(crl_ppo* h, const float* obs, const double* u, int32_t n, int32_t* action, float* logprob,
                       float* value) {
  CRL_GUARD_SETTLED(h);
  CRL_NEED_PARAMS(h, "crl_policy_act");
  if (n < 0 || (n > 0 && (!obs || !u || !action || !logprob))) { set_error("crl_policy_act: bad arguments"); return 1; }
  if (n == 0) return 0;
  const size_t d = (size_t)h->dc.D, N = (size_t)n;
  const size_t o_obs = 0, o_u = o_obs + ((N * d * 4 + 15) & ~(size_t)15), o_act = o_u + N * 8, o_lp = o_act + N * 4,
               o_val = o_lp + N * 4, total = o_val + N * 4;
  if (ensure_stage(h, total)) return 1;
  char* s = static_cast<char*>(h->stage);
  CRL_HIP_CHECK(hipMemcpyAsync(s + o_obs, obs, N * d * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(s + o_u, u, N * 8, hipMemcpyHostToDevice, h->stream));
  if (launch_policy_act(h, (const float*)(s + o_obs), (const double*)(s + o_u), n, (int32_t*)(s + o_act), (float*)(s + o_lp),
                        (float*)(s + o_val))) return 1;
  CRL_HIP_CHECK(hipMemcpyAsync(action, s + o_act, N * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(logprob, s + o_lp, N * 4, hipMemcpyDeviceToHost, h->stream));
  if (value) CRL_HIP_CHECK(hipMemcpyAsync(value, s + o_val, N * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int32_t crl_logprob_actions(crl_ppo* h, const float* obs, const int32_t* actions, int32_t n, float* logprob,
                            float* entropy) {
  CRL_GUARD_SETTLED(h);
  CRL_NEED_PARAMS(h, "crl_logprob_actions");
  if (n < 0 || (n > 0 && (!obs || !actions || !logprob || !entropy))) { set_error("crl_logprob_actions: bad arguments"); return 1; }
  if (n == 0) return 0;
  const size_t d = (size_t)h->dc.D, A = (size_t)h->dc.A, N = (size_t)n;
  const size_t o_obs = 0, o_act = o_obs + ((N * d * 4 + 15) & ~(size_t)15), o_lp = o_act + N * 4, o_ent = o_lp + N * 4,
               total = o_ent + N * A * 4;
  if (ensure_stage(h, total)) return 1;
  char* s = static_cast<char*>(h->stage);
  CRL_HIP_CHECK(hipMemcpyAsync(s + o_obs, obs, N * d * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(s + o_act, actions, N * 4, hipMemcpyHostToDevice, h->stream));
  if (launch_logprob_actions(h, (const float*)(s + o_obs), (const int32_t*)(s + o_act), n, (float*)(s + o_lp), (float*)(s + o_ent))) return 1;
  CRL_HIP_CHECK(hipMemcpyAsync(logprob, s + o_lp, N * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(entropy, s + o_ent, N * A * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int32_t crl_gae(int32_t device, const float* value, const float* reward, const uint8_t* terminal, const float* next_value,
                const uint8_t* next_done, int32_t nt, int32_t k, float gamma, float lambda, int32_t mode, float* adv,
                float* ret) {
  return crl_gae_opt(device, value, reward, terminal, next_value, next_done, nt, k, gamma, lambda, mode, adv, ret, 0, 0, 2);
}

int32_t crl_gae_opt(int32_t device, const float* value, const float* reward, const uint8_t* terminal, const float* next_value,
                    const uint8_t* next_done, int32_t nt, int32_t k, float gamma, float lambda, int32_t mode, float* adv,
                    float* ret, int32_t gae_seg, int32_t gae_tile, int32_t gae_nt_loads) {
  if (nt < 0 || k < 0) { set_error("crl_gae: negative size"); return 1; }
  if (!gae_seg_ok(gae_seg) || !gae_tile_ok(gae_tile) || (gae_seg == 4 && gae_tile > 4) || gae_nt_loads < 0 || gae_nt_loads > 2) {
    set_error("crl_gae_opt: gae_seg is 0 / 8 / 16 (4 with the streaming kernel), gae_tile 0 / 1 / 2 / 4 (streaming kernel) / 8 / 16 / 32 / 64 / 128 / 256, gae_nt_loads 0 / 1 / 2 (automatic)");
    return 1;
  }
  if (nt == 0 || k == 0) return 0;  // gae of an empty rollout is empty
  if (!value || !reward || !terminal || !adv) { set_error("crl_gae: null argument"); return 1; }
  if (mode == CRL_GAE_FIXED && (!next_value || !next_done)) { set_error("crl_gae: fixed mode needs next_value/next_done"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  const size_t B = (size_t)nt * k;
  char* buf = nullptr;
  const size_t o_v = 0, o_r = o_v + B * 4, o_a = o_r + B * 4, o_ret = o_a + B * 4, o_nv = o_ret + B * 4,
               o_t = o_nv + (size_t)nt * 4, o_nd = o_t + ((B + 15) & ~(size_t)15), total = o_nd + (size_t)nt;
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&buf), total));
  int rc = 0;
  auto fail = [&](hipError_t e, const char* what) { set_error(std::string(what) + ": " + hipGetErrorString(e)); rc = 1; };
  hipError_t e;
  if ((e = hipMemcpy(buf + o_v, value, B * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy value");
  if (!rc && (e = hipMemcpy(buf + o_r, reward, B * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy reward");
  if (!rc && (e = hipMemcpy(buf + o_t, terminal, B, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy terminal");
  if (!rc && next_value && (e = hipMemcpy(buf + o_nv, next_value, (size_t)nt * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy next_value");
  if (!rc && next_done && (e = hipMemcpy(buf + o_nd, next_done, (size_t)nt, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy next_done");
  if (!rc)
    rc = launch_gae(nullptr, (const float*)(buf + o_v), (const float*)(buf + o_r), (const uint8_t*)(buf + o_t),
                    next_value ? (const float*)(buf + o_nv) : nullptr, next_done ? (const uint8_t*)(buf + o_nd) : nullptr, nt, k,
                    gamma, lambda, mode, (float*)(buf + o_a), (float*)(buf + o_ret), nullptr, nullptr, gae_seg, gae_tile,
                    gae_nt_loads == 2 ? (B >= ((size_t)1 << 22) ? 1 : 0) : gae_nt_loads);   // automatic: inputs arrived by copies; streaming loads pay from ~4 M samples
  if (!rc && (e = hipDeviceSynchronize()) != hipSuccess) fail(e, "gae kernel");
  if (!rc && (e = hipMemcpy(adv, buf + o_a, B * 4, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy adv");
  if (!rc && ret && (e = hipMemcpy(ret, buf + o_ret, B * 4, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy ret");
  (void)hipFree(buf);
  return rc;
}

int32_t crl_rollout_store(crl_ppo* h, int32_t step, const float* obs, const int32_t* action, const float* logprob,
                          const float* reward, const uint8_t* terminal, const float* value) {
  CRL_GUARD_SETTLED(h);
  if (step < 0 || step >= h->dc.k) { set_error("crl_rollout_store: step out of range"); return 1; }
  if (!obs || !action || !logprob || !reward || !terminal || !value) { set_error("crl_rollout_store: null argument"); return 1; }
  const size_t nt = (size_t)h->dc.nt, d = (size_t)h->dc.D, off = nt * (size_t)step;
  CRL_HIP_CHECK(hipMemcpyAsync(h->obs + off * d, obs, nt * d * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->action + off, action, nt * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->logprob + off, logprob, nt * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->reward + off, reward, nt * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->terminal + off, terminal, nt, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->value + off, value, nt * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));  // host buffers are only borrowed for the call
  h->recs_dirty = true;
  return 0;
}

int32_t crl_env_reset(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  if (launch_env_reset(h)) return 1;
  h->env_ready = true;
  return 0;
}

static int ensure_env(crl_ppo* h) {  // ppo.jl:112-115 runs once before the loop
  if (h->env_ready) return 0;
  if (launch_env_reset(h)) return 1;
  h->env_ready = true;
  return 0;
}

int32_t crl_rollout_run(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  CRL_NEED_PARAMS(h, "crl_rollout_run");
  if (ensure_env(h)) return 1;
  h->recs_dirty = true;
  return launch_rollout(h);
}

int32_t crl_episode_stats_read(crl_ppo* h, crl_episode_stats* out) {
  CRL_GUARD(h);
  if (!out) { set_error("null argument"); return 1; }
  if (settle(h)) return 1;
  double v[4];
  CRL_HIP_CHECK(hipMemcpyAsync(v, h->ep_stats, sizeof(v), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  out->episodes = v[0]; out->return_sum = v[1]; out->length_sum = v[2]; out->return_max = v[3];
  return 0;
}

int32_t crl_episode_ring_enable(crl_ppo* h, int32_t capacity) {
  CRL_GUARD_SETTLED(h);
  if (capacity < 0 || capacity > (1 << 26)) { set_error("crl_episode_ring_enable: capacity must be in 0..2^26"); return 1; }
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (h->ep_ring) { (void)hipFree(h->ep_ring); h->ep_ring = nullptr; }
  if (h->ep_ring_count) { (void)hipFree(h->ep_ring_count); h->ep_ring_count = nullptr; }
  h->ep_ring_cap = 0;
  if (capacity == 0) return 0;
  if (dalloc(&h->ep_ring, (size_t)capacity) || dalloc(&h->ep_ring_count, 1)) return 1;
  h->ep_ring_cap = capacity;
  return 0;
}

int32_t crl_episode_ring_read(crl_ppo* h, crl_episode_record* out, int32_t max_records, int32_t* n_stored, int64_t* n_episodes) {
  CRL_GUARD(h);
  if (!n_stored || max_records < 0 || (max_records > 0 && !out)) { set_error("crl_episode_ring_read: bad arguments"); return 1; }
  *n_stored = 0;
  if (n_episodes) *n_episodes = 0;
  if (h->ep_ring_cap == 0) { set_error("crl_episode_ring_read: the ring is not enabled (crl_episode_ring_enable)"); return 1; }
  if (settle(h)) return 1;
  uint32_t cnt = 0;
  CRL_HIP_CHECK(hipMemcpyAsync(&cnt, h->ep_ring_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (n_episodes) *n_episodes = cnt;
  uint32_t n = cnt < (uint32_t)h->ep_ring_cap ? cnt : (uint32_t)h->ep_ring_cap;
  if (n > (uint32_t)max_records) n = (uint32_t)max_records;
  if (n) {
    CRL_HIP_CHECK(hipMemcpyAsync(out, h->ep_ring, sizeof(crl_episode_record) * n, hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  *n_stored = (int32_t)n;
  return 0;
}

static int compute_gae(crl_ppo* h) {
  h->recs_dirty = true;
  const bool fixed = h->cfg.gae_mode == CRL_GAE_FIXED;
  if (fixed && launch_next_value(h)) return 1;
  ProfScope ps(h, CRL_K_GAE, /*attach=*/true);
  return launch_gae(h->stream, h->value, h->reward, h->terminal, fixed ? h->next_value : nullptr, h->next_done, h->dc.nt,
                    h->dc.k, h->cfg.gamma, h->cfg.gae_lambda, h->cfg.gae_mode, h->adv, h->ret, ps.a, ps.b, (int)opt(h, OPT_GAE_SEG),
                    (int)opt(h, OPT_GAE_TILE), opt(h, OPT_GAE_NT_LOADS) == 2 ? ((h->cfg.env_kind == CRL_ENV_EXTERNAL && h->dc.B >= (1 << 22)) ? 1 : 0) : (int)opt(h, OPT_GAE_NT_LOADS));
}
int32_t crl_compute_gae(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  if (h->cfg.gae_mode == CRL_GAE_FIXED) CRL_NEED_PARAMS(h, "crl_compute_gae (fixed mode bootstraps from critic(next_obs))");
  return compute_gae(h);
}

// host_syncs: the caller synchronises with the stream anyway (crl_sync, statistics read-back); crl_shuffle does so only in the blocked
// Fisher-Yates mode (its overflow word), so the optimiser step's time-out word is not fetched from its other modes
static int check_bfy(crl_ppo* h, bool host_syncs) {
  if (peer_check(h)) return 1;
  if ((host_syncs || h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY) && fused_optim_check(h)) return 1;
  if (h->cfg.shuffle_mode != CRL_SHUFFLE_BLOCKED_FY) return 0;
  uint32_t err = 0;
  for (int z = 0; z < h->cfg.update_epochs && !err; ++z) {
    CRL_HIP_CHECK(hipMemcpyAsync(&err, h->bfy_ws + (size_t)z * (4 * 16384 + 8) + 3 * 16384 + 1, sizeof(err), hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  if (err) { set_error("blocked Fisher-Yates: a bucket overflowed its LDS leaf (probability < 1e-200; corrupted state?)"); return 1; }
  return 0;
}

int32_t crl_shuffle(crl_ppo* h, uint64_t epoch_id) {
  CRL_GUARD_SETTLED(h);
  h->slot_fresh &= ~(1u << h->cur_slot);
  if (launch_shuffle(h, epoch_id)) return 1;
  return check_bfy(h, /*host_syncs=*/false);
}

// local Σadv, Σadv² of the current slot's minibatches → adv_sums
static int adv_sums_local(crl_ppo* h) {
  if (h->wide) return launch_adv_stats_sums(h);
  h->slot_fresh &= ~(1u << h->cur_slot);   // recompute: the caller may have overwritten CRL_F_ADV_SUMS
  return ensure_records(h);
}

int32_t crl_adv_stats(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  if (adv_sums_local(h)) return 1;
  if (comm_allreduce(h, h->adv_sums, (size_t)h->dc.nmb * 2, true)) return 1;
  return launch_adv_stats_finish(h);
}

int32_t crl_adv_stats_local(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  return adv_sums_local(h);
}
int32_t crl_adv_stats_finish(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  return launch_adv_stats_finish(h);
}

static int update_step(crl_ppo* h, int mb, double eta, int apply, int slot, bool inline_fix = true) {
  // with a communicator attached and the optimiser following, the statistics ride in the optimiser launch (optim.hip);
  // the inline value-loss fix-up reads the flag the statistics raise, so it keeps them as their own launch
  // nothing between the gradient and the optimiser but — under data parallelism over the peer mailboxes — the exchange, which the launch runs itself
  // (no RCCL call, no host-side exchange, no inline value-loss fix-up): ONE launch reduces, exchanges, clips and steps (update.hip: reduce_optim_kernel)
  // (ranks that SHARE a GPU — functional runs on a 1-GPU box — each need their whole grid resident while they wait for one another's chunks: the
  // one-launch step is taken only while all of them fit with room to spare; otherwise the three-launch step, whose exchange kernel has no grid-wide wait)
  const long nb_step = (long)((h->P + 4 + 63) / 64);
  const bool local_or_peer = !has_comm(h) || (peer_active(h) && !h->comm && (long)peer_ranks_on_my_device(h) * nb_step * 10 <= h->fuse_optim_capacity * 6);
  const bool fused = apply && !h->wide && local_or_peer && !h->external_comm && !(inline_fix && h->cfg.clip_value_loss) && opt(h, OPT_FUSE_OPTIM) &&
                     h->fuse_optim_fits && (h->P & 63) <= 60;   // (the four loss sums ride behind the gradient in the last 64-float chunk; only the 4 / 2 / 64 shape reaches this path — P = 9,155, P mod 64 = 3 — so the last condition never decides; it documents the kernel's layout assumption)
  h->defer_stats = !fused && apply && !h->wide && has_comm(h) && !(inline_fix && h->cfg.clip_value_loss && h->world == 1);
  const int rc = launch_update(h, mb, h->stats_dev + slot, inline_fix, fused, eta);
  h->defer_stats = false;
  if (rc) return 1;
  if (apply && !fused && launch_optim(h, eta)) return 1;
  return 0;
}

int32_t crl_ppo_update_minibatch(crl_ppo* h, int32_t mb, double eta, int32_t apply_update, crl_ppo_stats* stats) {
  CRL_GUARD_SETTLED(h);
  CRL_NEED_PARAMS(h, "crl_ppo_update_minibatch");
  if (mb < 0 || mb >= h->dc.nmb) { set_error("crl_ppo_update_minibatch: minibatch index out of range"); return 1; }
  if (ensure_records(h)) return 1;
  if (update_step(h, mb, eta, apply_update, mb)) return 1;
  if (stats) {
    CRL_HIP_CHECK(hipMemcpyAsync(stats, h->stats_dev + mb, sizeof(crl_ppo_stats), hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Speculation guard (Q4). The fused kernels speculate on u = mean(v − R²) ≤ 0 (ppo.jl:232-237). Inside crl_ppo_iterate the
// exact fix-up is NOT enqueued per optimiser step (three early-exit launches on one GPU, two more collectives under RCCL):
// the common path stays speculative and a (global, sticky) device flag records a failed speculation. The flag is read back
// once per WINDOW of iterations (option guard_window, default 8) or whenever the host reads results — never per iteration. A window starts with a snapshot of everything an
// iteration mutates (parameters, Adam state, env state, episode accumulators); if the flag is up at the end, every rank
// restores the snapshot and repeats the window's iterations with the exact step. The flag derives from all-reduced sums,
// so all ranks take the same branch as long as they issue the same sequence of library calls.
// ---------------------------------------------------------------------------------------------------------------
static bool guard_on(const crl_ppo* h) { return !h->wide && h->cfg.clip_value_loss && !h->external_comm; }

struct EnvSnapLayout { size_t state, obs, t, done, ret, len, stats, ring, dscale, total; };
static EnvSnapLayout env_snap_layout(const crl_ppo* h) {
  const size_t nt = (size_t)h->dc.nt, d = (size_t)h->dc.D;
  EnvSnapLayout l;
  l.state = 0; l.obs = l.state + nt * d * 4; l.t = l.obs + nt * d * 4; l.ret = l.t + nt * 4; l.len = l.ret + nt * 4;
  l.stats = l.len + nt * 4; l.ring = l.stats + 32; l.dscale = l.ring + 16; l.done = l.dscale + 16; l.total = l.done + nt;
  return l;
}
static int guard_copy(crl_ppo* h, bool save) {
  const size_t P = (size_t)h->P, nt = (size_t)h->dc.nt, d = (size_t)h->dc.D;
  const EnvSnapLayout l = env_snap_layout(h);
  if (l.total > h->snap_env_bytes) { set_error("internal: env snapshot buffer too small"); return 1; }
  char* e = static_cast<char*>(h->snap_env);
  struct Pair { void* live; void* snap; size_t bytes; };
  const Pair pairs[] = {
      {h->params, h->snap, P * 4}, {h->adam_m, h->snap + P, P * 4}, {h->adam_v, h->snap + 2 * P, P * 4}, {h->betap, h->snap_betap, 24 * 8},
      {h->env_state, e + l.state, nt * d * 4}, {h->cur_obs, e + l.obs, nt * d * 4}, {h->env_t, e + l.t, nt * 4},
      {h->ep_return, e + l.ret, nt * 4}, {h->ep_length, e + l.len, nt * 4}, {h->ep_stats, e + l.stats, 32},
      {h->next_done, e + l.done, nt}, {h->ep_ring_count, e + l.ring, 4},
      // the sticky fp16x2 weight-gradient scale and the running largest |δ2| (mlp_x2.hpp): a replay starts from the scales the window
      // started with, so it is bit-identical to a run that never speculated
      {h->dscale, e + l.dscale, 16}};
  const void* src[16]; void* dst[16]; size_t bytes[16]; int n = 0;
  for (const Pair& p : pairs) {
    if (!p.live) continue;   // the episode ring is optional
    src[n] = save ? p.live : p.snap; dst[n] = save ? p.snap : p.live; bytes[n] = p.bytes; ++n;
  }
  return launch_guard_copy(h, src, dst, bytes, n);   // one launch (thirteen hipMemcpyAsync calls took 0.6 ms)
}

static double anneal_eta(const crl_ppo* h) {
  double eta = (double)h->cfg.lr;
  if (h->cfg.anneal_lr) {  // ppo.jl:118-121 (update is 1-based). The reference loop ends at num_updates; a caller that
    // keeps iterating past it gets eta = 0 rather than a negative step (gradient ascent)
    double frac = 1.0 - ((double)(h->iteration + 1) - 1.0) / (double)h->num_updates;
    if (frac < 0.0) frac = 0.0;
    eta = frac * (double)h->cfg.lr;
  }
  return eta;
}

// all update_epochs permutations of one iteration (ppo.jl:191-194) into the perm slots, on the handle's CURRENT stream
static int draw_epoch_permutations(crl_ppo* h, uint64_t ep0) {
  const int E = h->cfg.update_epochs;
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY) {   // one launch per pass for all epochs
    select_slot(h, 0);
    return launch_shuffle_epochs(h, ep0, E);
  }
  for (int ep = 0; ep < E; ++ep) {
    select_slot(h, ep);
    if (h->cfg.shuffle_mode == CRL_SHUFFLE_FISHER_YATES) {   // b_inds = shuffle(b_inds): each epoch shuffles the previous order
      if (ep == 0) { if (launch_iota(h)) return 1; }         // ppo.jl:191
      else CRL_HIP_CHECK(hipMemcpyAsync(h->perm, h->perm - h->dc.B, (size_t)h->dc.B * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    if (launch_shuffle(h, ep0 + (uint64_t)ep)) return 1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined read-back (crl_ppo_iterate_async / crl_ppo_drain). A host that logs every update (ppo.jl:147-165,246-248) reads, per iteration, the loss
// records, the episode statistics, the per-episode ring, the speculation flag and three error words. Read synchronously that is eight small copies with
// a stream synchronisation each and — worse — an idle GPU while the host wakes up and enqueues the next iteration's forty launches: 0.29 ms per iteration
// at 65536 envs (3 %), 0.18 ms at an 8192-env shard (9 %; scripts/readback_cost.py). Here the iteration's status is gathered by one launch into a device
// slot, copied to pinned host memory ON THE STREAM and fenced by an event; the host reads slot k one call later, after it has enqueued iteration k + 1.
// Slot layout (bytes): 0 sticky speculation flag (f64) | 8 optimiser time-out word | 12 peer time-out word | 16 ring count | 24 episode statistics (4 f64) |
// 56 blocked-shuffle overflow words (<= 8 epochs) | 128 loss records (update_epochs x num_minibatches x 64 B) | then the episode ring (capacity x 16 B).
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t ST_OFF_STICKY = 0, ST_OFF_OPTIM = 8, ST_OFF_PEER = 12, ST_OFF_RINGCNT = 16, ST_OFF_EP = 24, ST_OFF_BFY = 56, ST_OFF_STATS = 128;
constexpr int ST_MAX_EPOCHS = 8;
static size_t status_ring_off(const crl_ppo* h) { return (ST_OFF_STATS + sizeof(crl_ppo_stats) * (size_t)h->cfg.update_epochs * h->dc.nmb + 15) & ~(size_t)15; }
static int ensure_status(crl_ppo* h) {
  if (h->cfg.update_epochs > ST_MAX_EPOCHS && h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY) {
    set_error("crl_ppo_iterate_async: at most 8 update epochs with the blocked shuffle (its overflow words ride in the status slot)"); return 1;
  }
  if (h->status_dev[0] && h->status_ring_cap == h->ep_ring_cap) return 0;
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int s = 0; s < 2; ++s) {
    if (h->status_dev[s]) (void)hipFree(h->status_dev[s]);
    if (h->status_host[s]) (void)hipHostFree(h->status_host[s]);
    h->status_dev[s] = nullptr; h->status_host[s] = nullptr; h->status_iter[s] = -1;
  }
  h->status_bytes = status_ring_off(h) + sizeof(crl_episode_record) * (size_t)h->ep_ring_cap;
  for (int s = 0; s < 2; ++s) {
    CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->status_dev[s]), h->status_bytes));
    CRL_HIP_CHECK(hipMemset(h->status_dev[s], 0, h->status_bytes));
    CRL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->status_host[s]), h->status_bytes, hipHostMallocDefault));
    if (!h->status_ev[s]) CRL_HIP_CHECK(hipEventCreateWithFlags(&h->status_ev[s], hipEventDisableTiming));
  }
  h->status_ring_cap = h->ep_ring_cap;
  h->staged_last = -1; h->delivered_last = -1;
  return 0;
}
// the status of iteration k (just enqueued) into slot k & 1: one gather launch, one copy, one event — all in stream order, so they see iteration k's results and
// run before iteration k + 1's rollout clears the episode accumulators
static int stage_status(crl_ppo* h, int64_t k) {
  const int s = (int)(k & 1);
  char* d = h->status_dev[s];
  const void* src[16]; void* dst[16]; size_t bytes[16]; int n = 0;
  auto add = [&](const void* from, size_t off, size_t nb) { if (from && nb) { src[n] = from; dst[n] = d + off; bytes[n] = nb; ++n; } };
  add(h->vfix + 4, ST_OFF_STICKY, 8);
  add(h->ticket ? h->ticket + 1 : nullptr, ST_OFF_OPTIM, 4);
  add(peer_err_word(h), ST_OFF_PEER, 4);
  add(h->ep_ring_count, ST_OFF_RINGCNT, 4);
  add(h->ep_stats, ST_OFF_EP, 32);
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY)
    for (int z = 0; z < h->cfg.update_epochs; ++z) add(h->bfy_ws + (size_t)z * (4 * 16384 + 8) + 3 * 16384 + 1, ST_OFF_BFY + 4 * (size_t)z, 4);
  add(h->stats_dev, ST_OFF_STATS, sizeof(crl_ppo_stats) * (size_t)h->cfg.update_epochs * h->dc.nmb);
  if (h->ep_ring_cap > 0) add(h->ep_ring, status_ring_off(h), sizeof(crl_episode_record) * (size_t)h->ep_ring_cap);
  if (launch_guard_copy(h, src, dst, bytes, n)) return 1;
  CRL_HIP_CHECK(hipMemcpyAsync(h->status_host[s], d, h->status_bytes, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipEventRecord(h->status_ev[s], h->stream));
  h->status_iter[s] = k;
  h->staged_last = k;
  return 0;
}

// one pass of the ppo.jl:117-253 loop body
static int iterate_once(crl_ppo* h, bool exact) {
  const int E = h->cfg.update_epochs, nmb = h->dc.nmb;
  const double eta = anneal_eta(h);
  const uint64_t ep0 = (uint64_t)h->iteration * (uint64_t)E;
  h->recs_dirty = true;
  if (h->wide) {
    // layer-wise path: per-epoch shuffle → statistics (→ all-reduce) → optimiser steps, gathering through the permutation
    if (launch_rollout(h)) return 1;
    if (compute_gae(h)) return 1;
    if (h->cfg.shuffle_mode == CRL_SHUFFLE_FISHER_YATES && launch_iota(h)) return 1;  // ppo.jl:191
    for (int ep = 0; ep < E; ++ep) {
      if (launch_shuffle(h, ep0 + (uint64_t)ep, /*with_adv_sums=*/true)) return 1;
      if (launch_adv_stats_sums(h)) return 1;
      if (comm_allreduce(h, h->adv_sums, (size_t)nmb * 2, true)) return 1;
      if (launch_adv_stats_finish(h)) return 1;
      for (int mb = 0; mb < nmb; ++mb)
        if (update_step(h, mb, eta, 1, ep * nmb + mb)) return 1;
    }
    return 0;
  }
  // fused path. All update_epochs permutations are drawn up front — they depend on nothing the rollout or the optimiser
  // produces — on the second stream, next to the rollout kernel. One sequential pass over the advantages then leaves all E·nmb
  // advantage sums, which cross the ranks in ONE all-reduce per iteration; after that an optimiser step is
  // update (records fetched through the permutation) → reduce → (all-reduce) → Adam.
  const bool overlap = opt(h, OPT_SHUFFLE_OVERLAP) != 0;
  if (overlap) {
    CRL_HIP_CHECK(hipEventRecord(h->ev_fork, h->stream));          // the previous iteration's update kernels have read the slots
    CRL_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    std::swap(h->stream, h->stream2);
    const int rc = draw_epoch_permutations(h, ep0);
    std::swap(h->stream, h->stream2);
    if (rc) return 1;
    CRL_HIP_CHECK(hipEventRecord(h->ev_join, h->stream2));
  }
  const bool fuse = rollout_can_fuse_gae(h);
  if (launch_rollout(h, fuse)) return 1;
  if (!fuse && compute_gae(h)) return 1;
  if (launch_pack_records(h)) return 1;
  if (overlap) CRL_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  else if (draw_epoch_permutations(h, ep0)) return 1;
  if (launch_slot_adv_sums(h, 0, E)) return 1;
  {
    ProfScope ps(h, CRL_K_ADV_STATS);
    if (comm_allreduce(h, h->adv_sums_base, (size_t)E * nmb * 2, true)) return 1;
    if (launch_adv_stats_finish(h, 0, E)) return 1;
  }
  for (int ep = 0; ep < E; ++ep) {
    select_slot(h, ep);
    for (int mb = 0; mb < nmb; ++mb) {
      if (exact) {
        if (launch_update_exact_dp(h, mb, h->stats_dev + ep * nmb + mb)) return 1;
        if (launch_optim(h, eta)) return 1;
      } else if (update_step(h, mb, eta, 1, ep * nmb + mb, /*inline_fix=*/!guard_on(h))) return 1;
    }
  }
  return 0;   // the current slot stays at the last epoch: CRL_F_PERM reads back the b_inds the loop ended with
}

// Ends a guard window: reads the sticky flag (one host sync) and, if the speculation failed anywhere inside the window,
// restores its start and repeats its iterations exactly. No-op outside data parallelism or with an empty window.
static int settle(crl_ppo* h) {
  if (!guard_on(h) || h->window_count == 0) return 0;
  // a timed-out peer exchange leaves garbage in the all-reduced sums the sticky flag derives from: report it instead of replaying
  if (peer_check(h)) { h->window_count = 0; return 1; }
  double sticky = 0.0;
  CRL_HIP_CHECK(hipMemcpyAsync(&sticky, h->vfix + 4, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  const int n = h->window_count;
  h->window_count = 0;
  if (sticky == 0.0) return 0;
  if (guard_copy(h, /*save=*/false)) return 1;
  CRL_HIP_CHECK(hipMemsetAsync(h->vfix + 4, 0, sizeof(double), h->stream));
  h->iteration = h->snap_iteration;
  for (int i = 0; i < n; ++i) {
    if (iterate_once(h, /*exact=*/true)) return 1;
    h->iteration += 1;
    // pipelined read-back: the slots of the repeated iterations now hold what the exact pass produced (and a lowered flag)
    if (h->pipelined && h->status_dev[0] && stage_status(h, h->iteration - 1)) return 1;
  }
  CRL_HIP_CHECK(hipMemsetAsync(h->vfix + 4, 0, sizeof(double), h->stream));   // handled: lower the sticky flag
  h->exact_reruns += n;
  return 0;
}

int32_t crl_ppo_iterate(crl_ppo* h, int32_t n_iters, crl_ppo_stats* stats) {
  CRL_GUARD(h);
  CRL_NEED_PARAMS(h, "crl_ppo_iterate");
  if (h->cfg.env_kind == CRL_ENV_EXTERNAL) { set_error("crl_ppo_iterate needs an on-device env (CRL_ENV_CARTPOLE or CRL_ENV_SYNTHETIC)"); return 1; }
  const int E = h->cfg.update_epochs, nmb = h->dc.nmb;
  if (ensure_env(h)) return 1;
  const bool guard = guard_on(h);
  for (int it = 0; it < n_iters; ++it) {
    if (guard && h->window_count == 0) {
      if (guard_copy(h, /*save=*/true)) return 1;
      h->snap_iteration = h->iteration;
    }
    if (iterate_once(h, /*exact=*/false)) return 1;
    h->iteration += 1;
    if (guard && ++h->window_count >= h->window_len && settle(h)) return 1;
  }
  if (stats) {
    if (settle(h)) return 1;   // the host is about to look: make what it sees exact
    CRL_HIP_CHECK(hipMemcpyAsync(stats, h->stats_dev, sizeof(crl_ppo_stats) * (size_t)E * nmb, hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (check_bfy(h)) return 1;
  }
  if (h->world > 1 && h->external_comm && !h->wide && h->cfg.clip_value_loss) {
    // host-side exchange (crl_comm_init_external) cannot run the exact re-pass: fail loudly rather than train on a
    // speculative critic gradient
    double vf[8];
    CRL_HIP_CHECK(hipMemcpyAsync(vf, h->vfix, sizeof(vf), hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (vf[4] != 0.0) {
      set_error("value-loss branch u = mean(v - R^2) > 0 was taken under host-side gradient exchange; use crl_comm_init (RCCL) for the exact pass");
      return 1;
    }
  }
  return 0;
}

// hands the staged status of iteration k to the caller (see the block comment above stage_status)
static int deliver_status(crl_ppo* h, int64_t k, crl_ppo_iteration_report* rep, crl_ppo_stats* stats, crl_episode_record* ring, int32_t max_ring) {
  const int s = (int)(k & 1);
  if (h->status_iter[s] != k) { set_error("internal: the status slot does not hold the iteration asked for"); return 1; }
  CRL_HIP_CHECK(hipEventSynchronize(h->status_ev[s]));
  double sticky = 0.0;
  std::memcpy(&sticky, h->status_host[s] + ST_OFF_STICKY, 8);
  if (sticky != 0.0 && guard_on(h) && h->window_count > 0) {   // (the layer-wise path computes the exact value loss in line: its flag is informational)
    // a speculation failed inside the open guard window: the window is repeated exactly — every repeated iteration re-stages its slot — and the slot is read again
    if (settle(h)) return 1;
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->status_iter[s] != k) { set_error("internal: the replay did not re-stage the iteration asked for"); return 1; }
  }
  const char* p = h->status_host[s];
  uint32_t optim_err = 0, peer_err = 0, ring_cnt = 0;
  std::memcpy(&optim_err, p + ST_OFF_OPTIM, 4); std::memcpy(&peer_err, p + ST_OFF_PEER, 4); std::memcpy(&ring_cnt, p + ST_OFF_RINGCNT, 4);
  if (peer_err) { set_error("peer all-reduce timed out waiting for another rank (a rank died or the ranks issued different collectives)"); return 1; }
  if (optim_err) { set_error("reduce_optim_kernel: a block timed out at the grid meeting point (the grid was not fully resident, or the device is shared); set option fuse_optim = 0"); return 1; }
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY)
    for (int z = 0; z < h->cfg.update_epochs; ++z) {
      uint32_t e = 0;
      std::memcpy(&e, p + ST_OFF_BFY + 4 * (size_t)z, 4);
      if (e) { set_error("blocked Fisher-Yates: a bucket overflowed its LDS leaf (probability < 1e-200; corrupted state?)"); return 1; }
    }
  if (rep) {
    double ep[4];
    std::memcpy(ep, p + ST_OFF_EP, 32);
    rep->iteration = k;
    rep->episodes.episodes = ep[0]; rep->episodes.return_sum = ep[1]; rep->episodes.length_sum = ep[2]; rep->episodes.return_max = ep[3];
    rep->n_episodes = h->ep_ring_cap > 0 ? (int64_t)ring_cnt : (int64_t)ep[0];
    uint32_t n = h->ep_ring_cap > 0 ? (ring_cnt < (uint32_t)h->ep_ring_cap ? ring_cnt : (uint32_t)h->ep_ring_cap) : 0u;
    if (n > (uint32_t)(max_ring > 0 ? max_ring : 0)) n = (uint32_t)(max_ring > 0 ? max_ring : 0);
    rep->n_ring = (int32_t)n;
    if (n && ring) std::memcpy(ring, p + status_ring_off(h), sizeof(crl_episode_record) * n);
  }
  if (stats) std::memcpy(stats, p + ST_OFF_STATS, sizeof(crl_ppo_stats) * (size_t)h->cfg.update_epochs * h->dc.nmb);
  h->delivered_last = k;
  return 0;
}

int32_t crl_ppo_iterate_async(crl_ppo* h, crl_ppo_iteration_report* prev, crl_ppo_stats* prev_stats, crl_episode_record* prev_ring, int32_t max_ring) {
  CRL_GUARD(h);
  CRL_NEED_PARAMS(h, "crl_ppo_iterate_async");
  if (!prev) { set_error("crl_ppo_iterate_async: null report"); return 1; }
  if (max_ring < 0 || (max_ring > 0 && !prev_ring)) { set_error("crl_ppo_iterate_async: bad ring buffer"); return 1; }
  if (h->cfg.env_kind == CRL_ENV_EXTERNAL) { set_error("crl_ppo_iterate_async needs an on-device env (CRL_ENV_CARTPOLE or CRL_ENV_SYNTHETIC)"); return 1; }
  prev->iteration = -1; prev->n_ring = 0; prev->n_episodes = 0;
  prev->episodes.episodes = prev->episodes.return_sum = prev->episodes.length_sum = prev->episodes.return_max = 0.0;
  if (h->status_ring_cap != h->ep_ring_cap && h->staged_last > h->delivered_last) {
    set_error("crl_ppo_iterate_async: the episode ring was resized with an undelivered iteration pending (call crl_ppo_drain first)"); return 1;
  }
  if (ensure_status(h)) return 1;
  if (ensure_env(h)) return 1;
  h->pipelined = true;
  const bool guard = guard_on(h);
  if (guard && h->window_count == 0) {
    if (guard_copy(h, /*save=*/true)) return 1;
    h->snap_iteration = h->iteration;
  }
  if (iterate_once(h, /*exact=*/false)) return 1;
  h->iteration += 1;
  const int64_t k = h->iteration - 1;
  if (stage_status(h, k)) return 1;
  if (guard && ++h->window_count >= h->window_len && settle(h)) return 1;
  // the previous iteration's status: its copy finished long ago (iteration k is queued behind it), so this wait does not drain the GPU
  if (k - 1 > h->delivered_last && k >= 1 && h->status_iter[(k - 1) & 1] == k - 1)
    return deliver_status(h, k - 1, prev, prev_stats, prev_ring, max_ring);
  return 0;
}

int32_t crl_ppo_drain(crl_ppo* h, crl_ppo_iteration_report* last, crl_ppo_stats* last_stats, crl_episode_record* last_ring, int32_t max_ring) {
  CRL_GUARD(h);
  if (!last) { set_error("crl_ppo_drain: null report"); return 1; }
  if (max_ring < 0 || (max_ring > 0 && !last_ring)) { set_error("crl_ppo_drain: bad ring buffer"); return 1; }
  last->iteration = -1; last->n_ring = 0; last->n_episodes = 0;
  last->episodes.episodes = last->episodes.return_sum = last->episodes.length_sum = last->episodes.return_max = 0.0;
  if (!h->status_dev[0] || h->staged_last <= h->delivered_last) return 0;
  return deliver_status(h, h->staged_last, last, last_stats, last_ring, max_ring);
}

int32_t crl_ppo_exact_reruns(const crl_ppo* h, int64_t* n) {
  if (!h || !n) { set_error("null argument"); return 1; }
  *n = h->exact_reruns;
  return 0;
}

int32_t crl_ppo_iteration(const crl_ppo* h, int64_t* it) {
  if (!h || !it) { set_error("null argument"); return 1; }
  *it = h->iteration;
  return 0;
}

int32_t crl_comm_unique_id(uint8_t id[128]) { return comm_unique_id(id); }
int32_t crl_comm_info(char* path, size_t path_cap, int32_t* version) { int v = 0; const int rc = comm_info(path, path_cap, &v); if (version) *version = v; return rc; }

int32_t crl_comm_init(crl_ppo* h, const uint8_t id[128], int32_t world_size, int32_t rank) {
  CRL_GUARD_SETTLED(h);
  if (comm_init(h, id, world_size, rank)) return 1;
  if (reset_dw_scale(h)) return 1;
  // num_updates = total_timesteps ÷ (global batch) (ppo.jl:89-91)
  const int64_t gb = (int64_t)h->dc.B * h->world;
  h->num_updates = h->cfg.total_timesteps / gb;
  if (h->num_updates < 1) h->num_updates = 1;
  return 0;
}

int32_t crl_comm_peer_export(crl_ppo* h, int32_t world_size, int32_t rank, uint8_t handle[64]) {
  CRL_GUARD(h);
  if (!handle) { set_error("crl_comm_peer_export: null handle"); return 1; }
  if (settle(h)) return 1;
  return peer_export(h, world_size, rank, handle);
}

int32_t crl_comm_peer_attach(crl_ppo* h, const uint8_t* handles) {
  CRL_GUARD_SETTLED(h);
  if (!handles) { set_error("crl_comm_peer_attach: null handles"); return 1; }
  if (peer_attach(h, handles)) return 1;
  if (reset_dw_scale(h)) return 1;
  const int64_t gb = (int64_t)h->dc.B * h->world;
  h->num_updates = h->cfg.total_timesteps / gb;
  if (h->num_updates < 1) h->num_updates = 1;
  return 0;
}

int32_t crl_comm_init_external(crl_ppo* h, int32_t world_size, int32_t rank) {
  CRL_GUARD_SETTLED(h);
  if (world_size < 1 || rank < 0 || rank >= world_size) { set_error("crl_comm_init_external: bad world/rank"); return 1; }
  h->world = world_size; h->rank = rank; h->external_comm = true;
  if (reset_dw_scale(h)) return 1;
  const int64_t gb = (int64_t)h->dc.B * h->world;
  h->num_updates = h->cfg.total_timesteps / gb;
  if (h->num_updates < 1) h->num_updates = 1;
  return 0;
}

int32_t crl_comm_destroy(crl_ppo* h) {
  CRL_GUARD_SETTLED(h);
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  peer_destroy(h);
  comm_destroy(h);
  h->external_comm = false; h->world = 1; h->rank = 0;
  if (reset_dw_scale(h)) return 1;
  h->num_updates = h->cfg.total_timesteps / (int64_t)h->dc.B;
  if (h->num_updates < 1) h->num_updates = 1;
  return 0;
}

int32_t crl_ppo_set_option(crl_ppo* h, const char* key, int64_t value) {
  CRL_GUARD_SETTLED(h);   // an option may select a different kernel flavour: never inside an open guard window
  return opt_set(h, key, value);
}

int32_t crl_ppo_get_option(crl_ppo* h, const char* key, int64_t* value) {
  CRL_GUARD(h);
  if (!value) { set_error("crl_ppo_get_option: null argument"); return 1; }
  if (key && std::strcmp(key, "gemm_fallback_seen") == 0) {
    // read-only: 1 once a launch has run a role as bf16x3 because a hidden-layer weight left the fp16x2 window (|w| >= 255)
    double re = 0.0;
    if (h->wide) { *value = 0; return 0; }   // the layer-wise path scales its fp16x2 weight pieces per step: it has no fallback to take
    CRL_HIP_CHECK(hipMemcpyAsync(&re, h->vfix + 5, sizeof(re), hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
    *value = re != 0.0 ? 1 : 0;
    return 0;
  }
  const int id = opt_find(key);
  if (id < 0) { set_error(std::string("crl_ppo_get_option: unknown option '") + (key ? key : "(null)") + "'"); return 1; }
  *value = h->opt[id];
  return 0;
}

int32_t crl_ppo_option_count(int32_t* n) {
  if (!n) { set_error("null argument"); return 1; }
  *n = OPT_COUNT;
  return 0;
}
int32_t crl_ppo_option_name(int32_t index, const char** name, int64_t* dflt) {
  if (index < 0 || index >= OPT_COUNT || !name) { set_error("crl_ppo_option_name: index out of range"); return 1; }
  *name = kOpts[index].name;
  if (dflt) *dflt = kOpts[index].dflt;
  return 0;
}

int32_t crl_prof_enable(crl_ppo* h, int32_t on) {
  CRL_GUARD(h);
  h->prof = on < 0 ? 0 : (on > 2 ? 1 : on);
  return 0;
}
int32_t crl_prof_read(crl_ppo* h, int32_t kernel_id, double* total_ms, int64_t* launches) {
  CRL_GUARD(h);
  if (kernel_id < 0 || kernel_id >= CRL_K_COUNT) { set_error("bad kernel id"); return 1; }
  if (prof_collect(h)) return 1;
  if (total_ms) *total_ms = h->prof_slots[kernel_id].total_ms;
  if (launches) *launches = h->prof_slots[kernel_id].launches;
  return 0;
}
int32_t crl_prof_reset(crl_ppo* h) {
  CRL_GUARD(h);
  if (prof_collect(h)) return 1;
  for (int k = 0; k < CRL_K_COUNT; ++k) { h->prof_slots[k].total_ms = 0; h->prof_slots[k].launches = 0; }
  return 0;
}

}  // extern "C"
