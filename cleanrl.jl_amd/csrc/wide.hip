// wide.hip — the same PPO hot path (ppo.jl:21-45,123-166,197-251) for network shapes other than the specialised
// obs 4 / act 2 / 2×64 one: obs_dim ≤ 64, n_act ≤ 16, hidden ∈ {64, 128, 256} (BASELINE config C3: obs 8, act 4, 2×256).
//
// A 256-wide layer no longer fits the "whole network in registers + LDS" scheme of update.hip (W2 alone is 256 KB), so
// this path is layer-wise: activations of a whole minibatch live in HBM as (features, samples) column-major arrays —
// the reference's own layout — and every dense layer is one tiled v_mfma_f32_32x32x2_f32 GEMM:
//   forward          Y = act(W·X + b)            wide_dense_kernel<EPI_TANH | EPI_BIAS>
//   backward (data)  dX = (Wᵀ·dY) ⊙ (1 − X²)     wide_dense_kernel<EPI_DTANH> on a transposed weight copy
//   backward (weight) dW = dY·Xᵀ, db = Σ dY      wide_wgrad_kernel: the reduction runs over samples, both operands are
//                                                 read feature-fastest (no transpose anywhere), split over sample
//                                                 chunks into per-block partials, summed in fixed order afterwards
//   K ≤ 16 / N ≤ 16 weight gradients (dW1, dW3)  wide_skinny_kernel on the VALU (no MFMA shape fits)
// At 2×256 a layer is 131 KFLOP per sample against 2 KB of HBM traffic: the MFMA pipe, not HBM, bounds every GEMM.
// Because the critic forward is its own pass here, the value-loss scalar u = mean(v − R²) (ppo.jl:232, Q4) is known
// before any cotangent is formed: no speculation and no fix-up pass, and it is exact under data parallelism too.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "common.hpp"
#include "env.hpp"
#include "mlp_x2.hpp"
#include "mlp_x3.hpp"
#include "ppo_ctx.hpp"
#include "stats.hpp"

namespace crl {

constexpr int WXS = 36;    // LDS stride (floats) of one staged sample row: 32 k + 4 pad → conflict-free ds_read_b128
constexpr int WLS = 24;    // doubles per loss-kernel block partial: pg, Σ-entropy, Σ(v−R²), Σ value term, db3a[16], db3c
constexpr int AMAX = 16;
constexpr int AFUSE = 8;   // most head outputs the fused δ2 paths keep in registers (more ⇒ the separate launch)

enum { EPI_TANH = 0, EPI_BIAS = 1, EPI_DTANH = 2, EPI_STORE = 3 };

// ------------------------------------------------------------------------------------------------------
// Workspace
// ------------------------------------------------------------------------------------------------------
struct WideNetPack { int w1, w3, w2t, w3t, x3f, x3b, x2f, x2b, wmax, w1s, w1f, size; };  // float offsets of the packed weight copies of one network
constexpr int X3_SLAB_BF16 = 3 * 2 * 8 * 64 * 8;   // one 32-k slab of a 256-row matrix as bf16x3 A-fragments: [piece][kstep][ntile][lane][8]
constexpr int X2_SLAB_F16 = 2 * 2 * 8 * 64 * 8;    // the same slab as fp16x2 A-fragments of W·2^8 (mlp_x2.hpp)
static inline WideNetPack pack_layout(int H, int D8, int O8) {
  WideNetPack p;
  p.w1 = 0; p.w3 = p.w1 + H * D8; p.w2t = p.w3 + 32 * H; p.w3t = p.w2t + H * H;
  p.x3f = p.w3t + H * O8;
  const int x3 = H == 256 ? (H / 32) * X3_SLAB_BF16 / 2 : 0;   // floats
  p.x3b = p.x3f + x3;
  const int x2 = H == 256 ? (H / 32) * X2_SLAB_F16 / 2 : 0;    // floats
  p.x2f = p.x3b + x3; p.x2b = p.x2f + x2; p.wmax = p.x2b + x2; p.w1s = p.wmax + AMAX;
  p.w1f = p.w1s + (H == 256 ? H * D8 + H : 0);    // wide_fused.hpp: W1·2·log2(e) rows + bias, read through scalar loads
  p.size = p.w1f + (H == 256 ? 4096 + 256 : 0);   // … and W1 as fp16x2 A-fragments + bias for the producers' layer-1 product
  return p;
}

struct WideWs {
  int H, D, D8, A, A8, Mw;       // Mw = samples the activation buffers hold
  WideNetPack pk[2]; int pk_base[2];
  float* pack = nullptr;         // padded / transposed weight copies, rebuilt after every parameter change
  bool pack_dirty = true;
  float* wsc = nullptr;          // [2 networks][scale, 1/scale]: the power of two the fp16x2 pieces of W2 are staged with (wide_w2scale_kernel)
  float *h1[2] = {nullptr, nullptr}, *h2[2] = {nullptr, nullptr};  // [H × Mw] tanh activations, actor / critic
  float* z = nullptr;            // [A8 × Mw] logits, overwritten by their cotangent
  float* v = nullptr;            // [Mw] critic outputs
  float* dv8 = nullptr;          // [8 × Mw] critic-output cotangent in row 0 (rows 1-7 zero: K of the head GEMM is padded to 8)
  float *dA = nullptr, *dB = nullptr;  // [H × Mw] hidden-layer cotangents
  float* d2s = nullptr;                // [2][Mw] per-sample inverse scales of the split δ2 planes (option wide_d2_split)
  // gradient partials
  int S2 = 1, chunk2 = 32, Ss = 1, chunks = 256, nlb = 1;
  float *pW2[2] = {nullptr, nullptr}, *pB2[2] = {nullptr, nullptr};
  float *pW1[2] = {nullptr, nullptr}, *pB1[2] = {nullptr, nullptr};
  float* pW3[2] = {nullptr, nullptr};
  double* lpart = nullptr;       // [nlb][WLS]
  float* wrec = nullptr;         // [B][8]: action (bits), logprob, value, advantage | return, 0, 0, 0 — what the loss kernel reads of a sample, one 32-byte piece
  double* vpart = nullptr;       // [1024]
  double* u_dev = nullptr;       // [0] u (reserved), [1] = double(count) under DP
  int cus = 256;                 // compute units of the device (grid of the persistent fused backward)
  int w3_blocks = 0;             // 1: the last backward (wide_rs_bwd_kernel) left the dW3 partials itself (one per block, like dW1): no sweeps over h2
  int fb_blocks_net[2] = {0, 0}; // … per network (the register-stationary backward may split the CUs unevenly)
  int fb_blocks = 0;             // > 0: the last backward ran fused (wide_fused.hpp) and left this many dW1 / db1 partials per network
  int lds_max = 160 * 1024;      // LDS a block may ask for on this device: the fused kernels need 133-160 KB and are not chosen below that
};

static int walloc(float** p, size_t n) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float));
  if (e != hipSuccess) { set_error(std::string("hipMalloc (wide workspace): ") + hipGetErrorString(e)); return 1; }
  return 0;
}

bool wide_shape_ok(const crl_ppo_config* cfg, std::string* why) {
  const int H = cfg->hidden;
  if (H != 64 && H != 128 && H != 256) { *why = "hidden must be 64, 128 or 256"; return false; }
  if (cfg->obs_dim < 1 || cfg->obs_dim > 64) { *why = "obs_dim must be in 1..64"; return false; }
  if (cfg->n_act < 2 || cfg->n_act > AMAX) { *why = "n_act must be in 2..16"; return false; }
  return true;
}

void wide_destroy(crl_ppo* h) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (!w) return;
  void* ptrs[] = {w->pack, w->h1[0], w->h1[1], w->h2[0], w->h2[1], w->z, w->v, w->dv8, w->dA, w->dB, w->pW2[0], w->pW2[1],
                  w->pB2[0], w->pB2[1], w->pW1[0], w->pW1[1], w->pB1[0], w->pB1[1], w->pW3[0], w->pW3[1], w->lpart, w->wrec, w->vpart,
                  w->u_dev, w->wsc, w->d2s};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete w;
  h->wide_ws = nullptr;
}

int wide_create(crl_ppo* h) {
  WideWs* w = new (std::nothrow) WideWs();
  if (!w) { set_error("out of host memory"); return 1; }
  h->wide_ws = w;
  w->H = h->cfg.hidden; w->D = h->cfg.obs_dim; w->A = h->cfg.n_act;
  w->D8 = (w->D + 7) & ~7; w->A8 = (w->A + 7) & ~7;
  const int M = h->dc.M, nt = h->dc.nt;
  w->Mw = M > nt ? M : nt;
  { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cu > 0) w->cus = cu; }
  // gfx950 has 160 KB of LDS per workgroup; the #error guard of update.hip also admits gfx942 (64 KB), where the tile-resident kernels
  // cannot launch: there the layer-wise kernels (wide_fuse = 0, wide_rollout_persist = 1) run instead of a launch failure
  { int lds = 0; if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess && lds > 0) w->lds_max = lds; }
  w->pk[0] = pack_layout(w->H, w->D8, w->A8);
  w->pk[1] = pack_layout(w->H, w->D8, 8);
  w->pk_base[0] = 0; w->pk_base[1] = w->pk[0].size;
  const size_t H = (size_t)w->H, Mw = (size_t)w->Mw;
  int rc = 0;
  rc |= walloc(&w->pack, (size_t)w->pk[0].size + w->pk[1].size);
  rc |= walloc(&w->wsc, 8);   // [net][scale, 1/scale] of the W2 pieces, then the same of the W1 fragments (wide_fused.hpp)
  for (int n = 0; n < 2; ++n) { rc |= walloc(&w->h1[n], H * Mw); rc |= walloc(&w->h2[n], H * Mw); }
  rc |= walloc(&w->z, (size_t)w->A8 * Mw); rc |= walloc(&w->v, Mw); rc |= walloc(&w->dv8, 8 * Mw);
  rc |= walloc(&w->dA, H * Mw); rc |= walloc(&w->dB, H * Mw); rc |= walloc(&w->d2s, 2 * Mw);
  // weight-gradient splits: ≈2048-sample chunks for the MFMA kernel, 512-sample chunks for the VALU kernels
  // swept at C3 (M = 524,288; profiles/r02_c3_*): 4096-sample chunks 68.0 ms per iteration, 2048: 69.8, 8192: 75.9 — half the
  // partials to write and fold against one block per CU instead of two
  int ck = M >= 262144 ? 4096 : 2048;
  int s2 = (M + ck - 1) / ck; if (s2 > 512) s2 = 512; if (s2 < 1) s2 = 1;
  w->S2 = s2; w->chunk2 = (((M + s2 - 1) / s2) + 31) & ~31;
  int ss = (M + 511) / 512; if (ss > 1024) ss = 1024; if (ss < 1) ss = 1;
  w->Ss = ss; w->chunks = (M + ss - 1) / ss;
  w->nlb = (M + 255) / 256;
  for (int n = 0; n < 2; ++n) {
    const size_t NO = n ? 1 : (size_t)w->A;
    rc |= walloc(&w->pW2[n], (size_t)w->S2 * H * H); rc |= walloc(&w->pB2[n], (size_t)w->S2 * H);
    rc |= walloc(&w->pW1[n], (size_t)w->Ss * H * w->D); rc |= walloc(&w->pB1[n], (size_t)w->Ss * H);
    rc |= walloc(&w->pW3[n], (size_t)w->Ss * H * NO);
  }
  rc |= walloc(reinterpret_cast<float**>(&w->lpart), (size_t)w->nlb * WLS * 2);
  rc |= walloc(&w->wrec, (size_t)h->dc.B * 8);
  rc |= walloc(reinterpret_cast<float**>(&w->vpart), 1024 * 2);
  rc |= walloc(reinterpret_cast<float**>(&w->u_dev), 4 * 2);
  if (rc) { wide_destroy(h); return 1; }
  return 0;
}

bool wide_x2_active(const crl_ppo* h);
void wide_mark_params_changed(crl_ppo* h) {
  if (h->wide_ws) static_cast<WideWs*>(h->wide_ws)->pack_dirty = true;
}

// ------------------------------------------------------------------------------------------------------
// Weight packing: W1 → [H × D8] (zero columns beyond obs_dim), W3 → [32 × H] (zero rows beyond n_out),
// W2ᵀ [H × H], W3ᵀ → [H × O8]. W2 itself is used in place (it already is H × H column-major).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wide_pack_body(int bx, int nbx, const float* __restrict__ params, float* __restrict__ pack, int H, int D,
                                               int D8, int NO, int O8, int pbase, int kbase, WideNetPack pk) {
  const float* W1 = params + pbase;
  const float* W2 = W1 + H * D + H;
  const float* W3 = W2 + H * H + H;
  float* o = pack + kbase;
  const int total = pk.x3f;
  for (int i = bx * blockDim.x + threadIdx.x; i < total; i += nbx * blockDim.x) {
    float val;
    if (i < pk.w3) { const int n = i % H, k = i / H; val = k < D ? W1[n + H * k] : 0.0f; }
    else if (i < pk.w2t) { const int q = i - pk.w3; const int a = q & 31, k = q >> 5; val = a < NO ? W3[a + NO * k] : 0.0f; }
    else if (i < pk.w3t) { const int q = i - pk.w2t; const int k = q % H, n = q / H; val = W2[n + H * k]; }
    else { const int q = i - pk.w3t; const int k = q % H, a = q / H; val = a < NO ? W3[a + NO * k] : 0.0f; }
    o[i] = val;
  }
}
__global__ void __launch_bounds__(256) wide_pack_kernel(const float* __restrict__ params, float* __restrict__ pack, int H, int D,
                                                       int D8, int NO, int O8, int pbase, int kbase, WideNetPack pk) {
  wide_pack_body(blockIdx.x, gridDim.x, params, pack, H, D, D8, NO, O8, pbase, kbase, pk);
}

// bf16x3 A-fragments of W2 (dir 0: rows = outputs) and W2ᵀ (dir 1: rows = inputs) for wide_dense_x3_kernel:
// element e of lane l in (slab s, piece, kstep, ntile) is A[32·ntile + (l & 31)][32 s + 16 kstep + 8 (l >> 5) + e]
__global__ void __launch_bounds__(256) wide_pack_x3_kernel(const float* __restrict__ params, float* __restrict__ pack, int pbase,
                                                          int kbase, WideNetPack pk) {
  constexpr int H = 256;
  const int t = blockIdx.x * 256 + threadIdx.x;          // (dir, s, kstep, ntile, lane)
  if (t >= 2 * 8 * 2 * 8 * 64) return;
  const int lane = t & 63, ntile = (t >> 6) & 7, kstep = (t >> 9) & 1, sl = (t >> 10) & 7, dir = t >> 13;
  const int n = 32 * ntile + (lane & 31), k0 = 32 * sl + 16 * kstep + 8 * (lane >> 5);
  float x[8];
  const float* W = params + pbase;   // pbase = flat offset of this network's W2
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = dir ? W[(k0 + e) + H * n] : W[n + H * (k0 + e)];
  const P3 p3 = split3(x);
  __bf16* dst = reinterpret_cast<__bf16*>(pack + kbase + (dir ? pk.x3b : pk.x3f)) + (size_t)sl * X3_SLAB_BF16;
  const int fr = (kstep * 8 + ntile) * 64 + lane;        // fragment index inside one piece of the slab
  reinterpret_cast<bf16x8*>(dst)[0 * 1024 + fr] = p3.hi;
  reinterpret_cast<bf16x8*>(dst)[1 * 1024 + fr] = p3.mid;
  reinterpret_cast<bf16x8*>(dst)[2 * 1024 + fr] = p3.lo;
}

// The power of two the 256-wide fp16x2 weight pieces are staged with: largest |w|·scale lands in [2^14, 2^15), so W2 fits the fp16
// window whatever its magnitude (the fused 64-wide kernels use a fixed 2^8 and fall back to bf16x3 for |w| >= 255; round 2 raised an
// error here instead). A power of two is exact, and the GEMM epilogues take it back out of the f32 accumulator (DenseX3Args::wsc).
__device__ __forceinline__ void wide_w2scale_body(const float* __restrict__ W2, int n, float* __restrict__ wsc) {
  __shared__ float sm[16];
  float m = 0.0f;
  {   // 16-byte loads, eight in flight per thread (n = 65536: one block of 256 threads walks it in 8 rounds)
    const f32x4* W4 = reinterpret_cast<const f32x4*>(W2);
    const int n4 = (reinterpret_cast<size_t>(W2) & 15) ? 0 : (n >> 2), nt = blockDim.x;   // (the critic's W2 starts n_act floats further: scalar loads then)
    int i = threadIdx.x;
    for (; i + 7 * nt < n4; i += 8 * nt) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = W4[i + u * nt];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fabsf(v[u][e]));
    }
    for (; i < n4; i += nt) { const f32x4 v = W4[i]; for (int e = 0; e < 4; ++e) m = __builtin_fmaxf(m, __builtin_fabsf(v[e])); }
    for (int q = 4 * n4 + threadIdx.x; q < n; q += nt) m = __builtin_fmaxf(m, __builtin_fabsf(W2[q]));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = __builtin_fmaxf(m, sm[w]);
    int e = (int)((__float_as_uint(m) >> 23) & 0xFFu);          // biased exponent: m in [2^(e-127), 2^(e-126))
    if (!(m > 0.0f) || e >= 255) e = 127 + 6;                   // all zero / non-finite weights: 2^8 as in the fused kernels
    e = e < 40 ? 40 : (e > 220 ? 220 : e);
    wsc[0] = __uint_as_float((unsigned)(127 + 14 + 127 - e) << 23);   // 2^(14 − (e − 127)): m·scale in [2^14, 2^15)
    wsc[1] = __uint_as_float((unsigned)(e - 14) << 23);               // its inverse
  }
}
__global__ void __launch_bounds__(1024) wide_w2scale_kernel(const float* __restrict__ W2, int n, float* __restrict__ wsc) { wide_w2scale_body(W2, n, wsc); }
// fp16x2 A-fragments of W2·scale / W2ᵀ·scale (same fragment order as wide_pack_x3_kernel, two pieces)
__device__ __forceinline__ void wide_pack_x2_body(int bx, const float* __restrict__ params, float* __restrict__ pack, int pbase,
                                                  int kbase, WideNetPack pk, const float* __restrict__ wsc) {
  constexpr int H = 256;
  const int t = bx * 256 + threadIdx.x;          // (dir, s, kstep, ntile, lane)
  if (t >= 2 * 8 * 2 * 8 * 64) return;
  const int lane = t & 63, ntile = (t >> 6) & 7, kstep = (t >> 9) & 1, sl = (t >> 10) & 7, dir = t >> 13;
  const int n = 32 * ntile + (lane & 31), k0 = 32 * sl + 16 * kstep + 8 * (lane >> 5);
  float x[8];
  const float* W = params + pbase;
  const float scale = wsc[0];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float w = dir ? W[(k0 + e) + H * n] : W[n + H * (k0 + e)];
    x[e] = w * scale;
  }
  const P2 p2 = split2(x);
  _Float16* dst = reinterpret_cast<_Float16*>(pack + kbase + (dir ? pk.x2b : pk.x2f)) + (size_t)sl * X2_SLAB_F16;
  const int fr = (kstep * 8 + ntile) * 64 + lane;
  reinterpret_cast<f16x8*>(dst)[0 * 1024 + fr] = p2.hi;
  reinterpret_cast<f16x8*>(dst)[1 * 1024 + fr] = p2.lo;
}
__global__ void __launch_bounds__(256) wide_pack_x2_kernel(const float* __restrict__ params, float* __restrict__ pack, int pbase,
                                                          int kbase, WideNetPack pk, const float* __restrict__ wsc) {
  wide_pack_x2_body(blockIdx.x, params, pack, pbase, kbase, pk, wsc);
}
// wmax[a] = max_k |W3[a, k]|: with it Σ_a |δ3[a, m]|·wmax[a] bounds every |δ2[k, m]| of a sample — the per-sample (backward-
// data) and per-chunk (weight-gradient) fp16x2 scales come from this bound, no pass over δ2 needed
__device__ __forceinline__ void wide_wmax_body(const float* __restrict__ W3, int NO, int H, float* __restrict__ wmax) {   // one wave
  for (int a = 0; a < AMAX; ++a) {
    float m = 0.0f;
    if (a < NO) for (int k = threadIdx.x; k < H; k += 64) m = __builtin_fmaxf(m, __builtin_fabsf(W3[a + NO * k]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) wmax[a] = m;
  }
}
__global__ void __launch_bounds__(64) wide_wmax_kernel(const float* __restrict__ W3, int NO, int H, float* __restrict__ wmax) { wide_wmax_body(W3, NO, H, wmax); }

// option "wide_gemm": 2 (default) = 256-wide GEMMs as fp16x2 (three f16 MFMAs per product), 1 = bf16x3 (six; the fallback
// flavour without range limits), 0 = every GEMM on v_mfma_f32_32x32x2_f32
static bool wide_x3(const crl_ppo* h) { return opt(h, OPT_WIDE_GEMM) >= 1; }   // a split-product flavour (x2 or x3)
static bool wide_x2(const crl_ppo* h) { return opt(h, OPT_WIDE_GEMM) == 2; }
bool wide_x2_active(const crl_ppo* h) { return h->wide && h->cfg.hidden == 256 && wide_x2(h); }

__global__ void wide_pack_w1s_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int H, int D, int DP, float* __restrict__ out);
__global__ void wide_pack_w1f_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D, const float* __restrict__ w1sc, float* __restrict__ out);
__global__ void wide_w1scale_kernel(const float* __restrict__ W1, int n, float* __restrict__ w1sc);
// The 2x256 fp16x2 shape repacks seven images per network after every optimiser step; as fourteen launches of a few microseconds each that
// was 1.4 ms per C3 iteration. Two launches now (blockIdx.y = network, blockIdx.x selects the job): the scales and everything that does not
// need them, then the two packs that do.
struct PrepNet { int pbase; int kbase; WideNetPack pk; int NO; int O8; };
struct PrepArgs { const float* params; float* pack; float* wsc; PrepNet n[2]; int H, D, D8; int nb_pack, nb_w1s; };
__global__ void wide_prep_a_kernel(PrepArgs a);
__global__ void wide_prep_b_kernel(PrepArgs a);
static int ensure_pack(crl_ppo* h) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (!w->pack_dirty) return 0;
  if (w->H == 256 && wide_x2(h) && w->D <= 16) {
    PrepArgs a;
    a.params = h->params; a.pack = w->pack; a.wsc = w->wsc; a.H = w->H; a.D = w->D; a.D8 = w->D8;
    int nbp = 0;
    for (int n = 0; n < 2; ++n) {
      a.n[n].pbase = n ? (int)h->Pa : 0; a.n[n].kbase = w->pk_base[n]; a.n[n].pk = w->pk[n]; a.n[n].NO = n ? 1 : w->A; a.n[n].O8 = n ? 8 : w->A8;
      const int nb = (w->pk[n].x3f + 255) / 256;
      if (nb > nbp) nbp = nb;
    }
    a.nb_pack = nbp; a.nb_w1s = (w->H * w->D8 + w->H + 255) / 256;
    hipLaunchKernelGGL(wide_prep_a_kernel, dim3(a.nb_pack + a.nb_w1s + 3, 2), dim3(256), 0, h->stream, a);
    hipLaunchKernelGGL(wide_prep_b_kernel, dim3(64 + 2, 2), dim3(256), 0, h->stream, a);
    CRL_HIP_CHECK(hipGetLastError());
    w->pack_dirty = false;
    return 0;
  }
  for (int n = 0; n < 2; ++n) {
    const int NO = n ? 1 : w->A, O8 = n ? 8 : w->A8;
    hipLaunchKernelGGL(wide_pack_kernel, dim3((w->pk[n].x3f + 255) / 256), dim3(256), 0, h->stream, h->params, w->pack, w->H, w->D,
                       w->D8, NO, O8, n ? (int)h->Pa : 0, w->pk_base[n], w->pk[n]);
    if (w->H == 256) {
      if (!wide_x2(h))   // the bf16x3 fragments are read by the wide_gemm = 1 flavour only
        hipLaunchKernelGGL(wide_pack_x3_kernel, dim3(2 * 8 * 2 * 8 * 64 / 256), dim3(256), 0, h->stream, h->params, w->pack,
                           (n ? (int)h->Pa : 0) + w->H * w->D + w->H, w->pk_base[n], w->pk[n]);
      hipLaunchKernelGGL(wide_w2scale_kernel, dim3(1), dim3(1024), 0, h->stream, h->params + (n ? (int)h->Pa : 0) + w->H * w->D + w->H,
                         w->H * w->H, w->wsc + 2 * n);
      hipLaunchKernelGGL(wide_pack_x2_kernel, dim3(2 * 8 * 2 * 8 * 64 / 256), dim3(256), 0, h->stream, h->params, w->pack,
                         (n ? (int)h->Pa : 0) + w->H * w->D + w->H, w->pk_base[n], w->pk[n], w->wsc + 2 * n);
    }
    if (w->H == 256 && w->D <= 16) {
      const float* W1 = h->params + (n ? (int)h->Pa : 0);
      hipLaunchKernelGGL(wide_w1scale_kernel, dim3(1), dim3(256), 0, h->stream, W1, w->H * w->D, w->wsc + 4 + 2 * n);
      hipLaunchKernelGGL(wide_pack_w1f_kernel, dim3(2), dim3(256), 0, h->stream, W1, W1 + w->H * w->D, w->D, w->wsc + 4 + 2 * n,
                         w->pack + w->pk_base[n] + w->pk[n].w1f);
    }
    if (w->H == 256)
      hipLaunchKernelGGL(wide_pack_w1s_kernel, dim3((w->H * w->D8 + w->H + 255) / 256), dim3(256), 0, h->stream, h->params + (n ? (int)h->Pa : 0),
                         h->params + (n ? (int)h->Pa : 0) + w->H * w->D, w->H, w->D, w->D8, w->pack + w->pk_base[n] + w->pk[n].w1s);
    hipLaunchKernelGGL(wide_wmax_kernel, dim3(1), dim3(64), 0, h->stream,
                       h->params + (n ? (int)h->Pa : 0) + w->H * w->D + w->H + w->H * w->H + w->H, NO, w->H,
                       w->pack + w->pk_base[n] + w->pk[n].wmax);
  }
  CRL_HIP_CHECK(hipGetLastError());
  w->pack_dirty = false;
  return 0;
}

// Epilogue store of one 32×32 accumulator tile with full 128-B lines: the C fragment holds 4 consecutive rows per
// register quad but one SAMPLE per lane, so a direct store scatters 16-B pieces over 64 lines per instruction. The tile
// goes through a wave-private [32][36] LDS scratch instead and comes back row-major: 8 lanes write one sample's 32 rows
// (128 B), a wave instruction covers 8 whole lines. S (the stored tanh outputs) is read the same way.
// tanh of the layer-wise path: the reference's rational tanh_fast, or — where nothing is compared bit for bit (the update pass,
// the critic) — 1 − 2/(2^(2·log2(e)·x) + 1) on v_exp_f32 / v_rcp_f32 (mlp_x2.hpp: 5 instructions instead of 13, ≈1e-7 absolute)
__device__ __forceinline__ float wide_tanh(float x, bool fast) { return fast ? tanh_exp2(x, TWO_LOG2E, 1.0f) : tanh_fast(x); }

template <int EPI>
__device__ __forceinline__ void tile_out(float* scr, const f32x16& acc, int lane, int n0, int mbase, int M, const float* bias,
                                         const float* S, int lds, float* Y, int ldy, bool fast = false) {
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 o; o[0] = acc[4 * g]; o[1] = acc[4 * g + 1]; o[2] = acc[4 * g + 2]; o[3] = acc[4 * g + 3];
    *reinterpret_cast<f32x4*>(scr + j * 36 + 8 * g + 4 * hf) = o;
  }
  wave_lds_fence();
  const int c = lane & 7, n = n0 + 4 * c;
  f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
  if (EPI == EPI_TANH || EPI == EPI_BIAS) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = (lane >> 3) + 8 * it, gm = mbase + m;
    f32x4 v = *reinterpret_cast<const f32x4*>(scr + m * 36 + 4 * c);
    if (gm < M) {
      if (EPI == EPI_DTANH) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(S + (size_t)lds * gm + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.0f - sv[e] * sv[e]);
      } else if (EPI == EPI_TANH || EPI == EPI_BIAS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += bv[e]; if (EPI == EPI_TANH) v[e] = wide_tanh(v[e], fast); }
      }
      *reinterpret_cast<f32x4*>(Y + (size_t)ldy * gm + n) = v;
    }
  }
  wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------------
// Dense layer: Y[n, m] = epi( Σ_k W[n, k]·X[k, m] ).  256 threads; the block owns NP = 32·WN·TN output rows (all of
// them) × MB = 32·WM·TM samples; K is walked in slabs of 32 staged through LDS (W slab: a contiguous copy; X slab:
// one 128-B row piece per sample). Three blocks fit a CU, so one block's staging hides under the others' MFMAs.
// k-pairing: MFMA step c of an 8-k group takes k = 8j+c from lanes 0-31 and k = 8j+4+c from lanes 32-63, so every lane
// reads its B operand as one ds_read_b128.
// ------------------------------------------------------------------------------------------------------
struct DenseArgs {
  const float* W; int Kp;                                  // packed [NP × Kp] column-major, Kp % 8 == 0
  const float* X; int ldx; int Kt; const int32_t* idx;     // sample m's features at X + ldx·(idx ? idx[m] : m), Kt valid
  const float* bias; const float* S; int lds;              // bias[Nt]; S: stored tanh outputs at the Y positions
  float* Y; int ldy; int Nt; int M;
  int fast_act = 0;   // EPI_TANH: the exp2-based activation of mlp_x2.hpp instead of tanh_fast (update path and critic only)
};

template <int WN, int TN, int WM, int TM, int EPI>
__device__ __forceinline__ void wide_dense_body(const DenseArgs& a) {
  constexpr int NP = 32 * WN * TN, MB = 32 * WM * TM;
  static_assert(WN * WM == 4, "four waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;            // [32][NP]
  float* Xl = smem + 32 * NP;  // [MB][WXS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int m0 = blockIdx.x * MB;
  f32x16 acc[TN][TM];
#pragma unroll
  for (int x = 0; x < TN; ++x)
#pragma unroll
    for (int y = 0; y < TM; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  const bool xfast = ((a.ldx & 3) == 0) && ((a.Kt & 3) == 0);

  // software pipeline: slab s+1 travels global → registers while slab s is multiplied out of LDS
  constexpr int WR = NP / 32, XR = MB / 32;   // float4 per thread per slab
  f32x4 wr[WR], xr[XR];
  auto fetch = [&](int k0) {
    const int ks = (a.Kp - k0) < 32 ? (a.Kp - k0) : 32;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.W + (size_t)NP * k0);
    const int n4 = NP * ks / 4;
#pragma unroll
    for (int u = 0; u < WR; ++u) { const int i = tid + 256 * u; if (i < n4) wr[u] = src[i]; }
    const int q4 = ks >> 2, n = MB * q4;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + 256 * u;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (i < n) {
        const int mm = i / q4, q = i - mm * q4, m = m0 + mm, k = k0 + 4 * q;
        if (m < a.M && k < a.Kt) {
          const size_t row = a.idx ? (size_t)a.idx[m] : (size_t)m;
          const float* p = a.X + row * (size_t)a.ldx + k;
          if (xfast) v = *reinterpret_cast<const f32x4*>(p);
          else {
            v[0] = p[0];
            if (k + 1 < a.Kt) v[1] = p[1];
            if (k + 2 < a.Kt) v[2] = p[2];
            if (k + 3 < a.Kt) v[3] = p[3];
          }
        }
      }
      xr[u] = v;
    }
  };
  auto stash = [&](int k0) {
    const int ks = (a.Kp - k0) < 32 ? (a.Kp - k0) : 32;
    f32x4* dst = reinterpret_cast<f32x4*>(Wl);
    const int n4 = NP * ks / 4;
#pragma unroll
    for (int u = 0; u < WR; ++u) { const int i = tid + 256 * u; if (i < n4) dst[i] = wr[u]; }
    const int q4 = ks >> 2, n = MB * q4;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + 256 * u;
      if (i < n) { const int mm = i / q4, q = i - mm * q4; *reinterpret_cast<f32x4*>(Xl + mm * WXS + 4 * q) = xr[u]; }
    }
  };

  if (xfast && (a.Kp & 31) == 0 && a.Kt == a.Kp) {
    // whole 32-k slabs of an aligned activation array (every hidden-layer GEMM): per-thread source pointers and LDS
    // slots are fixed for the whole K walk — one pointer bump per load, no index arithmetic in the loop
    const f32x4* wsrc[WR]; const f32x4* xsrc[XR]; bool xok[XR];
#pragma unroll
    for (int u = 0; u < WR; ++u) wsrc[u] = reinterpret_cast<const f32x4*>(a.W) + tid + 256 * u;
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + 256 * u, mm = i >> 3, q = i & 7, m = m0 + mm;
      xok[u] = m < a.M;
      const size_t row = xok[u] ? (a.idx ? (size_t)a.idx[m] : (size_t)m) : 0;
      xsrc[u] = reinterpret_cast<const f32x4*>(a.X + row * (size_t)a.ldx) + q;
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(Wl) + tid;
    float* xdst = Xl + (tid >> 3) * WXS + 4 * (tid & 7);      // slot of u = 0; u adds 32 rows
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int nslab = a.Kp >> 5;
#pragma unroll
    for (int u = 0; u < WR; ++u) wr[u] = wsrc[u][0];
#pragma unroll
    for (int u = 0; u < XR; ++u) xr[u] = xok[u] ? xsrc[u][0] : zero4;
    for (int sl = 0; sl < nslab; ++sl) {
      if (sl) __syncthreads();
#pragma unroll
      for (int u = 0; u < WR; ++u) wdst[256 * u] = wr[u];
#pragma unroll
      for (int u = 0; u < XR; ++u) *reinterpret_cast<f32x4*>(xdst + 32 * u * WXS) = xr[u];
      __syncthreads();
      if (sl + 1 < nslab) {
#pragma unroll
        for (int u = 0; u < WR; ++u) wr[u] = wsrc[u][(size_t)(sl + 1) * (NP * 8)];
#pragma unroll
        for (int u = 0; u < XR; ++u) xr[u] = xok[u] ? xsrc[u][(sl + 1) * 8] : zero4;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 b[TM];
#pragma unroll
        for (int y = 0; y < TM; ++y) b[y] = *reinterpret_cast<const f32x4*>(Xl + ((wm * TM + y) * 32 + j) * WXS + 8 * jj + 4 * hf);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int x = 0; x < TN; ++x) {
            const float av = Wl[(8 * jj + 4 * hf + c) * NP + (wn * TN + x) * 32 + j];
#pragma unroll
            for (int y = 0; y < TM; ++y) acc[x][y] = mfma32(av, b[y][c], acc[x][y]);
          }
        }
      }
    }
  } else {
    fetch(0);
    stash(0);
    __syncthreads();
    for (int k0 = 0; k0 < a.Kp; k0 += 32) {
      const int ks = (a.Kp - k0) < 32 ? (a.Kp - k0) : 32;
      const bool more = k0 + 32 < a.Kp;
      if (more) fetch(k0 + 32);
      for (int jj = 0; jj < (ks >> 3); ++jj) {
        f32x4 b[TM];
#pragma unroll
        for (int y = 0; y < TM; ++y) b[y] = *reinterpret_cast<const f32x4*>(Xl + ((wm * TM + y) * 32 + j) * WXS + 8 * jj + 4 * hf);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int x = 0; x < TN; ++x) {
            const float av = Wl[(8 * jj + 4 * hf + c) * NP + (wn * TN + x) * 32 + j];
#pragma unroll
            for (int y = 0; y < TM; ++y) acc[x][y] = mfma32(av, b[y][c], acc[x][y]);
          }
        }
      }
      if (more) {
        __syncthreads();
        stash(k0 + 32);
        __syncthreads();
      }
    }
  }

  const bool yfast = ((a.ldy & 3) == 0) && ((a.Nt & 3) == 0) && (EPI != EPI_DTANH || (a.lds & 3) == 0);
  if (yfast && a.Nt == NP) {
    // every row tile is complete: line-coalesced stores through a wave-private LDS scratch (the operand slabs are dead)
    __syncthreads();
    float* scr = smem + wave * (32 * 36);
#pragma unroll
    for (int x = 0; x < TN; ++x)
#pragma unroll
      for (int y = 0; y < TM; ++y)
        tile_out<EPI>(scr, acc[x][y], lane, (wn * TN + x) * 32, m0 + (wm * TM + y) * 32, a.M, a.bias, a.S, a.lds, a.Y, a.ldy, a.fast_act != 0);
    return;
  }
#pragma unroll
  for (int x = 0; x < TN; ++x) {
#pragma unroll
    for (int y = 0; y < TM; ++y) {
      const int m = m0 + (wm * TM + y) * 32 + j;
      if (m >= a.M) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (wn * TN + x) * 32 + 8 * g + 4 * hf;   // rows n..n+3 = registers 4g..4g+3 (rowmap)
        if (n >= a.Nt) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= a.Nt) continue;
          float o = acc[x][y][4 * g + e];
          if (EPI == EPI_DTANH) { const float sv = a.S[(size_t)a.lds * m + n + e]; o = o * (1.0f - sv * sv); }
          else { o += a.bias[n + e]; if (EPI == EPI_TANH) o = wide_tanh(o, a.fast_act != 0); }
          a.Y[(size_t)a.ldy * m + n + e] = o;
        }
      }
    }
  }
}

template <int WN, int TN, int WM, int TM, int EPI>
__global__ void __launch_bounds__(256) wide_dense_kernel(DenseArgs a) { wide_dense_body<WN, TN, WM, TM, EPI>(a); }
// the same layer of BOTH networks in one launch (blockIdx.y = network): the rollout's per-step forward passes are too small to
// fill the chip one network at a time
template <int WN, int TN, int WM, int TM, int EPI>
__global__ void __launch_bounds__(256) wide_dense_pair_kernel(DenseArgs a0, DenseArgs a1) {
  if (blockIdx.y == 0) wide_dense_body<WN, TN, WM, TM, EPI>(a0); else wide_dense_body<WN, TN, WM, TM, EPI>(a1);
}

// ------------------------------------------------------------------------------------------------------
// The same dense layer for the 256×256 hidden GEMMs on the bf16 matrix pipe at f32 accuracy (bf16x3, mlp_x3.hpp):
// W comes pre-split into A-fragments (wide_pack_x3_kernel, once per optimiser step) and is copied slab by slab into LDS
// as it stands; activations are split into their three bf16 pieces while they are stashed into LDS (each element once
// per block, reused by all 8 row tiles). 48 v_mfma_f32_32x32x16_bf16 per wave and slab (1,536 matrix-pipe cycles that
// overlap the VALU) replace 64 v_mfma_f32_32x32x2_f32 (4,096 cycles that do not).
// ------------------------------------------------------------------------------------------------------
struct DenseX3Args {
  const float* Wx3;                      // [K/32 slabs][piece][kstep][ntile][lane][8] bf16
  const float* X; int K;                 // X: [K × M] column-major, ld = K = 256
  const float* bias; const float* S;     // epilogue operands (ld 256)
  float* Y; int M;
  // optional fused head (EPI_TANH, 8-wave variant): Z[a, m] = Σ_n W3[a, n]·Y[n, m] + b3[a] while the tile is at hand
  const float* W3t; const float* b3; float* Z; int A; int ldz;   // W3t: [256 × ·] column-major (ld 256); Z null = no head
  // optional virtual input (EPI_DTANH): X holds h2 and the operand is formed on the fly as δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²) from
  // the head cotangent dZ[·, m] (ld ldd, zero-padded) — the [256 × M] δ2 array is never written or read
  const float* dZ; int ldd; int Ad;      // dZ null = X is the operand itself; Ad = live rows of dZ
  // fp16x2 backward-data (wide_dense_x2_kernel<EPI_DTANH>): the head cotangent and wmax, from which each sample's scale comes
  const float* bz; int bld; int bA; const float* wmax;
  int fast_act = 0;   // as DenseArgs::fast_act
  const float* wsc = nullptr;   // fp16x2 kernels: {scale, 1/scale} of this network's W2 pieces (wide_w2scale_kernel)
};
constexpr int X3ROW = 40;                // bf16 per staged sample row per piece: 32 k + 8 pad (80 B: conflict-free b128)

__device__ __forceinline__ void split3x4(const f32x4 v, uint2& h, uint2& m, uint2& l) {
  const uint32_t h0 = cvt_pk_bf16(v[0], v[1]), h1 = cvt_pk_bf16(v[2], v[3]);
  const float r0 = v[0] - __uint_as_float(h0 << 16), r1 = v[1] - __uint_as_float(h0 & 0xFFFF0000u);
  const float r2 = v[2] - __uint_as_float(h1 << 16), r3 = v[3] - __uint_as_float(h1 & 0xFFFF0000u);
  const uint32_t m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
  const float s0 = r0 - __uint_as_float(m0 << 16), s1 = r1 - __uint_as_float(m0 & 0xFFFF0000u);
  const float s2 = r2 - __uint_as_float(m1 << 16), s3 = r3 - __uint_as_float(m1 & 0xFFFF0000u);
  h = make_uint2(h0, h1); m = make_uint2(m0, m1); l = make_uint2(cvt_pk_bf16(s0, s1), cvt_pk_bf16(s2, s3));
}

// tanh epilogue that also accumulates the head's partial dot products of this 32-row tile while the values still sit in
// the C fragment (lane = sample, registers = rows): hp[m_local][a] += Σ_{n in tile} W3[a, n]·tanh(acc + b)[n, m]. One
// cross-half exchange per output instead of a lane reduction; the tile is then stored line-coalesced by tile_out.
__device__ __forceinline__ void tile_tanh_head(float* scr, f32x16 acc, int lane, int n0, int mloc0, int mbase, int M,
                                               const float* bias, float* Y, const float* W3t, int A, float* hp, int hs, float cs = 1.0f,
                                               bool fast = false) {
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + 8 * g + 4 * hf);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * g + e] = wide_tanh(__builtin_fmaf(acc[4 * g + e], cs, bv[e]), fast);
  }
  for (int aa = 0; aa < A; ++aa) {
    float p = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W3t + (size_t)256 * aa + n0 + 8 * g + 4 * hf);
#pragma unroll
      for (int e = 0; e < 4; ++e) p = __builtin_fmaf(w[e], acc[4 * g + e], p);
    }
    p += xor32(p);
    if (hf == 0) hp[(mloc0 + j) * hs + aa] += p;
  }
  tile_out<EPI_STORE>(scr, acc, lane, n0, mbase, M, nullptr, nullptr, 0, Y, 256);
}

template <int EPI, int TM, int NW>
__global__ void __launch_bounds__(64 * NW) wide_dense_x3_kernel(DenseX3Args a) {
  // NW waves share one W slab; a wave owns TN = 8/NW row tiles × TM sample tiles
  constexpr int NT = 64 * NW, TN = 8 / NW, MB = 32 * TM;
  constexpr int WR = X3_SLAB_BF16 * 2 / 16 / NT;               // 16-B pieces of the W slab per thread (12 or 6)
  constexpr int XR = (MB * 8 + NT - 1) / NT;                   // float4 pieces of the X slab per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  __bf16* Wl = reinterpret_cast<__bf16*>(smx);                 // one slab of A-fragments (48 KB)
  __bf16* Xl = Wl + X3_SLAB_BF16;                              // [piece][MB][X3ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int m0 = blockIdx.x * MB;
  f32x16 acc[TN][TM];
#pragma unroll
  for (int x = 0; x < TN; ++x)
#pragma unroll
    for (int y = 0; y < TM; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  const u32x4v* wsrc = reinterpret_cast<const u32x4v*>(a.Wx3) + tid;
  const f32x4* xsrc[XR]; bool xok[XR];
#pragma unroll
  for (int u = 0; u < XR; ++u) {
    const int i = tid + NT * u, mm = i >> 3, q = i & 7, m = m0 + mm;
    xok[u] = (i < MB * 8) && (m < a.M);
    xsrc[u] = reinterpret_cast<const f32x4*>(a.X + (size_t)(xok[u] ? m : 0) * a.K) + q;
  }
  u32x4v wr[WR]; f32x4 xr[XR];
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  // virtual input: this thread's samples are fixed for the whole K walk, so their head cotangents sit in registers
  const bool virt = (EPI == EPI_DTANH) && a.dZ != nullptr;   // compile-time dead in the forward instantiations
  float dzr[XR][AFUSE];
  if (virt) {
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + NT * u, mm = i >> 3, m = m0 + mm;
#pragma unroll
      for (int q2 = 0; q2 < AFUSE; ++q2) dzr[u][q2] = (xok[u] && q2 < a.Ad) ? a.dZ[(size_t)a.ldd * m + q2] : 0.0f;
    }
  }
  auto load_x = [&](int sl, int u) -> f32x4 {
    if (!xok[u]) return zero4;
    f32x4 v = xsrc[u][sl * 8];
    if (virt) {
      const int k = 32 * sl + 4 * ((tid + NT * u) & 7);
      f32x4 sacc = zero4;
#pragma unroll
      for (int q2 = 0; q2 < AFUSE; ++q2) {
        if (q2 < a.Ad) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(a.W3t + (size_t)256 * q2 + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[e] = __builtin_fmaf(w[e], dzr[u][q2], sacc[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = sacc[e] * (1.0f - v[e] * v[e]);
    }
    return v;
  };
#pragma unroll
  for (int u = 0; u < WR; ++u) wr[u] = wsrc[NT * u];
#pragma unroll
  for (int u = 0; u < XR; ++u) xr[u] = load_x(0, u);
  const int nslab = a.K >> 5;
  for (int sl = 0; sl < nslab; ++sl) {
    if (sl) __syncthreads();
#pragma unroll
    for (int u = 0; u < WR; ++u) reinterpret_cast<u32x4v*>(Wl)[tid + NT * u] = wr[u];
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + NT * u, mm = i >> 3, q = i & 7;
      if (i < MB * 8) {
        uint2 h, m, l;
        split3x4(xr[u], h, m, l);
        *reinterpret_cast<uint2*>(Xl + (0 * MB + mm) * X3ROW + 4 * q) = h;
        *reinterpret_cast<uint2*>(Xl + (1 * MB + mm) * X3ROW + 4 * q) = m;
        *reinterpret_cast<uint2*>(Xl + (2 * MB + mm) * X3ROW + 4 * q) = l;
      }
    }
    __syncthreads();
    if (sl + 1 < nslab) {
#pragma unroll
      for (int u = 0; u < WR; ++u) wr[u] = wsrc[(size_t)(sl + 1) * (X3_SLAB_BF16 / 8) + NT * u];
#pragma unroll
      for (int u = 0; u < XR; ++u) xr[u] = load_x(sl + 1, u);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      P3 af[TN], bf[TM];
#pragma unroll
      for (int x = 0; x < TN; ++x) {
        const int fr = (ks * 8 + TN * wave + x) * 64 + lane;
        af[x].hi = reinterpret_cast<const bf16x8*>(Wl)[0 * 1024 + fr];
        af[x].mid = reinterpret_cast<const bf16x8*>(Wl)[1 * 1024 + fr];
        af[x].lo = reinterpret_cast<const bf16x8*>(Wl)[2 * 1024 + fr];
      }
#pragma unroll
      for (int y = 0; y < TM; ++y) {
        const int off = (32 * y + j) * X3ROW + 16 * ks + 8 * hf;
        bf[y].hi = *reinterpret_cast<const bf16x8*>(Xl + 0 * MB * X3ROW + off);
        bf[y].mid = *reinterpret_cast<const bf16x8*>(Xl + 1 * MB * X3ROW + off);
        bf[y].lo = *reinterpret_cast<const bf16x8*>(Xl + 2 * MB * X3ROW + off);
      }
#pragma unroll
      for (int x = 0; x < TN; ++x)
#pragma unroll
        for (int y = 0; y < TM; ++y) acc[x][y] = mfma_x3(af[x], bf[y], acc[x][y]);
    }
  }
  __syncthreads();
  float* scr = reinterpret_cast<float*>(smx) + wave * (32 * 36);
  if (EPI == EPI_TANH && NW == 8 && a.Z) {
    // fused head: per-wave partials over its 32 rows, then a fixed-order fold over the 8 waves
    const int hs = a.ldz;
    float* hp_all = reinterpret_cast<float*>(smx) + NW * (32 * 36);
    float* hp = hp_all + wave * (MB * hs);
    for (int i = lane; i < MB * hs; i += 64) hp[i] = 0.0f;
    wave_lds_fence();
#pragma unroll
    for (int y = 0; y < TM; ++y)
      tile_tanh_head(scr, acc[0][y], lane, wave * 32, 32 * y, m0 + 32 * y, a.M, a.bias, a.Y, a.W3t, a.A, hp, hs, 1.0f, a.fast_act != 0);
    __syncthreads();
    for (int i = tid; i < MB * a.A; i += NT) {
      const int m = i / a.A, aa = i - m * a.A;
      float z = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) z += hp_all[w8 * (MB * hs) + m * hs + aa];
      if (m0 + m < a.M) a.Z[(size_t)a.ldz * (m0 + m) + aa] = z + a.b3[aa];
    }
    return;
  }
#pragma unroll
  for (int x = 0; x < TN; ++x)
#pragma unroll
    for (int y = 0; y < TM; ++y)
      tile_out<EPI>(scr, acc[x][y], lane, (TN * wave + x) * 32, m0 + 32 * y, a.M, a.bias, a.S, 256, a.Y, 256, a.fast_act != 0);
}

template <int EPI>
static int dense_x3_launch(hipStream_t st, const DenseX3Args& a) {
  if (a.M <= 0) return 0;
  const size_t head32 = a.Z ? (size_t)8 * 32 * 36 * 4 + (size_t)8 * 32 * a.ldz * 4 : 0;
  const size_t head64 = a.Z ? (size_t)8 * 32 * 36 * 4 + (size_t)8 * 64 * a.ldz * 4 : 0;
  if (a.M <= 32768) {
    size_t smem = X3_SLAB_BF16 * 2 + 3 * 32 * X3ROW * 2;
    if (head32 > smem) smem = head32;
    hipLaunchKernelGGL((wide_dense_x3_kernel<EPI, 1, 8>), dim3((a.M + 31) / 32), dim3(512), smem, st, a);
  } else {
    size_t smem = X3_SLAB_BF16 * 2 + 3 * 64 * X3ROW * 2;
    if (head64 > smem) smem = head64;
    hipLaunchKernelGGL((wide_dense_x3_kernel<EPI, 2, 8>), dim3((a.M + 63) / 64), dim3(512), smem, st, a);
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// The 256×256 hidden GEMMs as fp16x2 (mlp_x2.hpp): the structure of wide_dense_x3_kernel with two half-precision pieces
// per operand and three MFMAs per product. Scales are exact powers of two:
//   forward (EPI_TANH):      X = h1 ∈ (−1, 1) scaled by 2^14 while it is split; the accumulator holds 2^22·(W·h1) and the
//                            epilogue multiplies by 2^-22 before the bias and the tanh;
//   backward-data (EPI_DTANH): X = δ2 — each sample (column) scaled by 2^(14 − ⌈log2 bound⌉) with
//                            bound = Σ_a |δ3[a, m]|·max_k |W3[a, k]| ≥ max_k |δ2[k, m]| (no pass over δ2); the epilogue
//                            applies the inverse (and the weights' 2^-8) per sample.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split2x4(const f32x4 v, float s, uint2& h, uint2& l) {
  f32x2 a, b; a[0] = v[0] * s; a[1] = v[1] * s; b[0] = v[2] * s; b[1] = v[3] * s;
  const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
  // x − float(hi) as ONE v_fma_mix_f32 per value (as split2 of mlp_x2.hpp: the compiler's own form is v_cvt_f32_f16 + v_sub_f32)
  f32x2 ra, rb;
  const uint32_t hab = __builtin_bit_cast(uint32_t, ha), hbb = __builtin_bit_cast(uint32_t, hb);
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra[0]) : "v"(hab), "v"(a[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ra[1]) : "v"(hab), "v"(a[1]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rb[0]) : "v"(hbb), "v"(b[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb[1]) : "v"(hbb), "v"(b[1]));
  const f16x2 la = __builtin_convertvector(ra, f16x2), lb = __builtin_convertvector(rb, f16x2);
  h = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  l = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}
// 2^(14 − e), 2^(e − 14) for bound in [2^e, 2^(e+1)) (bound = 0 or tiny: any scale will do)
__device__ __forceinline__ void pow2_scale(float bound, float& s, float& inv) {
  int e = (int)((__float_as_uint(bound) >> 23) & 0xFFu);
  e = e < 16 ? 16 : (e > 250 ? 250 : e);
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 14) << 23);
}
// one accumulator tile through the wave's LDS scratch and out with whole 128-B lines (tile_out) with the fp16x2 unscale:
// EPI_TANH: tanh(acc·cs + bias); EPI_DTANH: acc·inv[sample]·cs·(1 − S²)
template <int EPI>
__device__ __forceinline__ void tile_out_x2(float* scr, const f32x16& acc, int lane, int n0, int mloc0, int mbase, int M, const float* bias,
                                            const float* S, float* Y, float cs, const float* inv_lds, bool fast = false) {
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 o; o[0] = acc[4 * g]; o[1] = acc[4 * g + 1]; o[2] = acc[4 * g + 2]; o[3] = acc[4 * g + 3];
    *reinterpret_cast<f32x4*>(scr + j * 36 + 8 * g + 4 * hf) = o;
  }
  wave_lds_fence();
  const int c = lane & 7, n = n0 + 4 * c;
  f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
  if (EPI == EPI_TANH) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int m = (lane >> 3) + 8 * it, gm = mbase + m;
    f32x4 v = *reinterpret_cast<const f32x4*>(scr + m * 36 + 4 * c);
    if (gm < M) {
      if (EPI == EPI_DTANH) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(S + (size_t)256 * gm + n);
        const float f = inv_lds[mloc0 + m] * cs;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] * f) * (1.0f - sv[e] * sv[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wide_tanh(__builtin_fmaf(v[e], cs, bv[e]), fast);
      }
      *reinterpret_cast<f32x4*>(Y + (size_t)256 * gm + n) = v;
    }
  }
  wave_lds_fence();
}

}  // namespace crl
#include "wide_fused.hpp"
namespace crl {

template <int EPI, int TM>
__device__ __forceinline__ void wide_dense_x2_body(const DenseX3Args& a) {
  constexpr int NW = 8, NT = 512, MB = 32 * TM;
  constexpr int WR = X2_SLAB_F16 * 2 / 16 / NT;                // 16-B pieces of the W slab per thread (4)
  constexpr int XR = (MB * 8 + NT - 1) / NT;                   // float4 pieces of the X slab per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  _Float16* Wl = reinterpret_cast<_Float16*>(smx);             // one slab of A-fragments (32 KB)
  _Float16* Xl = Wl + X2_SLAB_F16;                             // [piece][MB][X3ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int m0 = blockIdx.x * MB;
  // [MB] scale | [MB] inverse of the block's samples, behind everything the staging loop and the epilogue use
  constexpr int SC_OFF = (X2_SLAB_F16 * 2 + 2 * MB * X3ROW * 2 > NW * 32 * 36 * 4 ? X2_SLAB_F16 * 2 + 2 * MB * X3ROW * 2 : NW * 32 * 36 * 4);
  float* sc = reinterpret_cast<float*>(smx + SC_OFF);
  if (EPI == EPI_DTANH) {
    for (int t = tid; t < MB; t += NT) {
      const int m = m0 + t;
      float bound = 0.0f;
      if (m < a.M)
        for (int q2 = 0; q2 < a.bA; ++q2) bound = __builtin_fmaf(__builtin_fabsf(a.bz[(size_t)a.bld * m + q2]), a.wmax[q2], bound);
      float s1, i1;
      pow2_scale(bound, s1, i1);
      sc[t] = s1; sc[MB + t] = i1;
    }
    __syncthreads();
  }
  f32x16 acc[TM];
#pragma unroll
  for (int y = 0; y < TM; ++y)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[y][r] = 0.0f;
  const u32x4v* wsrc = reinterpret_cast<const u32x4v*>(a.Wx3) + tid;
  const f32x4* xsrc[XR]; bool xok[XR]; float xs[XR];
#pragma unroll
  for (int u = 0; u < XR; ++u) {
    const int i = tid + NT * u, mm = i >> 3, q = i & 7, m = m0 + mm;
    xok[u] = (i < MB * 8) && (m < a.M);
    xsrc[u] = reinterpret_cast<const f32x4*>(a.X + (size_t)(xok[u] ? m : 0) * a.K) + q;
    xs[u] = EPI == EPI_DTANH ? sc[i < MB * 8 ? mm : 0] : X2_ACT_SCALE;
  }
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  const int nslab = a.K >> 5;
  // one slab: registers → LDS (split on the way), barrier, refill the registers with slab `next` (if any), multiply out of LDS
  auto slab = [&](u32x4v (&wr)[WR], f32x4 (&xr)[XR], int sl, int next) {
    if (sl) __syncthreads();
#pragma unroll
    for (int u = 0; u < WR; ++u) reinterpret_cast<u32x4v*>(Wl)[tid + NT * u] = wr[u];
#pragma unroll
    for (int u = 0; u < XR; ++u) {
      const int i = tid + NT * u, mm = i >> 3, q = i & 7;
      if (i < MB * 8) {
        uint2 hh, ll;
        split2x4(xr[u], xs[u], hh, ll);
        *reinterpret_cast<uint2*>(Xl + (0 * MB + mm) * X3ROW + 4 * q) = hh;
        *reinterpret_cast<uint2*>(Xl + (1 * MB + mm) * X3ROW + 4 * q) = ll;
      }
    }
    __syncthreads();
    if (next < nslab) {
#pragma unroll
      for (int u = 0; u < WR; ++u) wr[u] = wsrc[(size_t)next * (X2_SLAB_F16 / 8) + NT * u];
#pragma unroll
      for (int u = 0; u < XR; ++u) xr[u] = xok[u] ? xsrc[u][next * 8] : zero4;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      P2 af, bf[TM];
      const int fr = (ks * 8 + wave) * 64 + lane;
      af.hi = reinterpret_cast<const f16x8*>(Wl)[0 * 1024 + fr];
      af.lo = reinterpret_cast<const f16x8*>(Wl)[1 * 1024 + fr];
#pragma unroll
      for (int y = 0; y < TM; ++y) {
        const int off = (32 * y + j) * X3ROW + 16 * ks + 8 * hf;
        bf[y].hi = *reinterpret_cast<const f16x8*>(Xl + 0 * MB * X3ROW + off);
        bf[y].lo = *reinterpret_cast<const f16x8*>(Xl + 1 * MB * X3ROW + off);
      }
#pragma unroll
      for (int y = 0; y < TM; ++y) acc[y] = mfma_x2(af, bf[y], acc[y]);
    }
  };
  auto fetch = [&](u32x4v (&wr)[WR], f32x4 (&xr)[XR], int sl) {
#pragma unroll
    for (int u = 0; u < WR; ++u) wr[u] = wsrc[(size_t)sl * (X2_SLAB_F16 / 8) + NT * u];
#pragma unroll
    for (int u = 0; u < XR; ++u) xr[u] = xok[u] ? xsrc[u][sl * 8] : zero4;
  };
  u32x4v wr[WR]; f32x4 xr[XR];
  fetch(wr, xr, 0);
  for (int sl = 0; sl < nslab; ++sl) slab(wr, xr, sl, sl + 1);
  __syncthreads();
  float* scr = reinterpret_cast<float*>(smx) + wave * (32 * 36);
  if (EPI == EPI_TANH && a.Z) {
    // fused head: per-wave partials over its 32 rows, then a fixed-order fold over the 8 waves
    const int hs = a.ldz;
    float* hp_all = reinterpret_cast<float*>(smx) + NW * (32 * 36);
    float* hp = hp_all + wave * (MB * hs);
    for (int i = lane; i < MB * hs; i += 64) hp[i] = 0.0f;
    wave_lds_fence();
#pragma unroll
    for (int y = 0; y < TM; ++y)
      tile_tanh_head(scr, acc[y], lane, wave * 32, 32 * y, m0 + 32 * y, a.M, a.bias, a.Y, a.W3t, a.A, hp, hs, a.wsc[1] * (1.0f / X2_ACT_SCALE), a.fast_act != 0);
    __syncthreads();
    for (int i = tid; i < MB * a.A; i += NT) {
      const int m = i / a.A, aa = i - m * a.A;
      float z = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) z += hp_all[w8 * (MB * hs) + m * hs + aa];
      if (m0 + m < a.M) a.Z[(size_t)a.ldz * (m0 + m) + aa] = z + a.b3[aa];
    }
    return;
  }
#pragma unroll
  for (int y = 0; y < TM; ++y)
    tile_out_x2<EPI>(scr, acc[y], lane, wave * 32, 32 * y, m0 + 32 * y, a.M, a.bias, a.S, a.Y,
                     EPI == EPI_TANH ? a.wsc[1] * (1.0f / X2_ACT_SCALE) : a.wsc[1], sc + MB, a.fast_act != 0);
}

template <int EPI, int TM>
__global__ void __launch_bounds__(512) wide_dense_x2_kernel(DenseX3Args a) { wide_dense_x2_body<EPI, TM>(a); }
template <int EPI, int TM>
__global__ void __launch_bounds__(512) wide_dense_x2_pair_kernel(DenseX3Args a0, DenseX3Args a1) {
  if (blockIdx.y == 0) wide_dense_x2_body<EPI, TM>(a0); else wide_dense_x2_body<EPI, TM>(a1);
}

static size_t dense_x2_smem(const DenseX3Args& a, int MB) {
  size_t region = (size_t)X2_SLAB_F16 * 2 + (size_t)2 * MB * X3ROW * 2;
  const size_t epi = (size_t)8 * 32 * 36 * 4;
  if (epi > region) region = epi;
  size_t smem = region + (size_t)2 * MB * 4;                              // + per-sample scale / inverse
  const size_t head = a.Z ? epi + (size_t)8 * MB * a.ldz * 4 : 0;         // fused head partials (forward only: no scales needed)
  return head > smem ? head : smem;
}

template <int EPI>
static int dense_x2_launch(hipStream_t st, const DenseX3Args& a) {
  if (a.M <= 0) return 0;
  // samples per block: 32 (rollout-sized batches: the grid still covers the chip), 64, or 128 for minibatches — every block
  // streams all of W2 (256 KB as fp16x2 fragments), so a larger block halves that traffic and doubles the MFMAs per staged slab
  const int MB = a.M > 131072 ? 128 : a.M > 32768 ? 64 : 32;
  const size_t smem = dense_x2_smem(a, MB);
  if (MB == 128) hipLaunchKernelGGL((wide_dense_x2_kernel<EPI, 4>), dim3((a.M + 127) / 128), dim3(512), smem, st, a);
  else if (MB == 64) hipLaunchKernelGGL((wide_dense_x2_kernel<EPI, 2>), dim3((a.M + 63) / 64), dim3(512), smem, st, a);
  else hipLaunchKernelGGL((wide_dense_x2_kernel<EPI, 1>), dim3((a.M + 31) / 32), dim3(512), smem, st, a);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int EPI>
static int dense_launch(hipStream_t st, int NP, const DenseArgs& a) {
  if (a.M <= 0) return 0;
  switch (NP) {
    case 256:
      // small batches (rollout: M = num_envs): 32-sample tiles so that at least two blocks land on every CU
      if (a.M <= 32768) hipLaunchKernelGGL((wide_dense_kernel<4, 2, 1, 1, EPI>), dim3((a.M + 31) / 32), dim3(256), sizeof(float) * (32 * 256 + 32 * WXS), st, a);
      else hipLaunchKernelGGL((wide_dense_kernel<4, 2, 1, 2, EPI>), dim3((a.M + 63) / 64), dim3(256), sizeof(float) * (32 * 256 + 64 * WXS), st, a);
      break;
    case 128: hipLaunchKernelGGL((wide_dense_kernel<4, 1, 1, 2, EPI>), dim3((a.M + 63) / 64), dim3(256), sizeof(float) * (32 * 128 + 64 * WXS), st, a); break;
    case 64: hipLaunchKernelGGL((wide_dense_kernel<2, 1, 2, 2, EPI>), dim3((a.M + 127) / 128), dim3(256), sizeof(float) * (32 * 64 + 128 * WXS), st, a); break;
    case 32: hipLaunchKernelGGL((wide_dense_kernel<1, 1, 4, 2, EPI>), dim3((a.M + 255) / 256), dim3(256), sizeof(float) * (32 * 32 + 256 * WXS), st, a); break;
    default: set_error("wide path: unsupported layer width"); return 1;
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// flat Flux parameter offsets inside one network
struct NetOff { int W1, b1, W2, b2, W3, b3, size; };
static NetOff net_off(int H, int D, int NO) {
  NetOff o; o.W1 = 0; o.b1 = H * D; o.W2 = o.b1 + H; o.b2 = o.W2 + H * H; o.W3 = o.b2 + H; o.b3 = o.W3 + NO * H; o.size = o.b3 + NO;
  return o;
}

// forward of one network over M samples: h1, h2 kept in the workspace, head output to out (ld ldo)
// fast_act: the exp2-based activation (wide_tanh) — the update pass and the critic; the actor of the rollout / get_action keeps
// tanh_fast because its logits decide action indices that are compared bit for bit
static int wide_forward(crl_ppo* h, int net, const float* X, int ldx, const int32_t* idx, int M, float* out, int ldo, bool fast_act = false) {
  fast_act = fast_act && !opt(h, OPT_WIDE_TANH_RATIONAL);
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  const int H = w->H, NO = net ? 1 : w->A;
  const NetOff o = net_off(H, w->D, NO);
  const float* P = h->params + (net ? h->Pa : 0);
  const float* pk = w->pack + w->pk_base[net];
  DenseArgs a;
  a.idx = idx; a.S = nullptr; a.lds = 0; a.M = M; a.fast_act = fast_act ? 1 : 0;
  a.W = pk + w->pk[net].w1; a.Kp = w->D8; a.X = X; a.ldx = ldx; a.Kt = w->D; a.bias = P + o.b1; a.Y = w->h1[net]; a.ldy = H; a.Nt = H;
  if (dense_launch<EPI_TANH>(h->stream, H, a)) return 1;
  a.idx = nullptr;
  if (H == 256 && wide_x3(h)) {
    DenseX3Args x;
    x.Wx3 = pk + w->pk[net].x3f; x.X = w->h1[net]; x.K = H; x.bias = P + o.b2; x.S = nullptr; x.Y = w->h2[net]; x.M = M;
    const bool fuse = true;   // the head comes out of the layer-2 epilogue (tile_tanh_head)
    x.W3t = pk + w->pk[net].w3t; x.b3 = P + o.b3; x.Z = fuse ? out : nullptr; x.A = NO; x.ldz = ldo;
    x.dZ = nullptr; x.ldd = 0; x.Ad = 0; x.bz = nullptr; x.bld = 0; x.bA = 0; x.wmax = nullptr; x.fast_act = fast_act ? 1 : 0;
    if (wide_x2(h)) { x.Wx3 = pk + w->pk[net].x2f; x.wsc = w->wsc + 2 * net; if (dense_x2_launch<EPI_TANH>(h->stream, x)) return 1; }
    else if (dense_x3_launch<EPI_TANH>(h->stream, x)) return 1;
    if (fuse) return 0;   // the head came out of the layer-2 epilogue
  } else {
    a.W = P + o.W2; a.Kp = H; a.X = w->h1[net]; a.ldx = H; a.Kt = H; a.bias = P + o.b2; a.Y = w->h2[net];
    if (dense_launch<EPI_TANH>(h->stream, H, a)) return 1;
  }
  a.ldx = H; a.Kt = H;
  a.W = pk + w->pk[net].w3; a.Kp = H; a.X = w->h2[net]; a.bias = P + o.b3; a.Y = out; a.ldy = ldo; a.Nt = NO;
  return dense_launch<EPI_BIAS>(h->stream, 32, a);
}

// Rollout step: actor and critic forward on the same observations with each layer of the two networks in ONE launch
// (blockIdx.y = network). Only for the 2×256 fp16x2 configuration with the fused head and M ≤ 32768 (the shapes the pair kernels
// are instantiated for); anything else runs the two networks one after the other.
static int wide_forward_pair(crl_ppo* h, const float* X, int ldx, int M, float* outA, int ldoA, float* outC, int ldoC) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  const bool fast_ok = !opt(h, OPT_WIDE_TANH_RATIONAL);
  if (!(w->H == 256 && wide_x2(h) && M > 0 && M <= 32768)) {
    if (wide_forward(h, 0, X, ldx, nullptr, M, outA, ldoA)) return 1;          // ppo.jl:127
    return wide_forward(h, 1, X, ldx, nullptr, M, outC, ldoC, true);           // ppo.jl:128
  }
  const int H = 256;
  DenseArgs a[2]; DenseX3Args x[2];
  for (int net = 0; net < 2; ++net) {
    const int NO = net ? 1 : w->A;
    const NetOff o = net_off(H, w->D, NO);
    const float* P = h->params + (net ? h->Pa : 0);
    const float* pk = w->pack + w->pk_base[net];
    const int fast = (net == 1 && fast_ok) ? 1 : 0;      // the actor keeps tanh_fast: its logits decide bit-compared action indices
    a[net].idx = nullptr; a[net].S = nullptr; a[net].lds = 0; a[net].M = M; a[net].fast_act = fast;
    a[net].W = pk + w->pk[net].w1; a[net].Kp = w->D8; a[net].X = X; a[net].ldx = ldx; a[net].Kt = w->D; a[net].bias = P + o.b1;
    a[net].Y = w->h1[net]; a[net].ldy = H; a[net].Nt = H;
    x[net].Wx3 = pk + w->pk[net].x2f; x[net].X = w->h1[net]; x[net].K = H; x[net].bias = P + o.b2; x[net].S = nullptr; x[net].Y = w->h2[net]; x[net].M = M;
    x[net].W3t = pk + w->pk[net].w3t; x[net].b3 = P + o.b3; x[net].Z = net ? outC : outA; x[net].A = NO; x[net].ldz = net ? ldoC : ldoA;
    x[net].dZ = nullptr; x[net].ldd = 0; x[net].Ad = 0; x[net].bz = nullptr; x[net].bld = 0; x[net].bA = 0; x[net].wmax = nullptr; x[net].fast_act = fast;
    x[net].wsc = w->wsc + 2 * net;
  }
  hipLaunchKernelGGL((wide_dense_pair_kernel<4, 2, 1, 1, EPI_TANH>), dim3((M + 31) / 32, 2), dim3(256), sizeof(float) * (32 * 256 + 32 * WXS), h->stream, a[0], a[1]);
  const size_t s0 = dense_x2_smem(x[0], 32), s1 = dense_x2_smem(x[1], 32);
  hipLaunchKernelGGL((wide_dense_x2_pair_kernel<EPI_TANH, 1>), dim3((M + 31) / 32, 2), dim3(512), s0 > s1 ? s0 : s1, h->stream, x[0], x[1]);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Weight gradient of a hidden layer: dW[n, k] = Σ_m dY[n, m]·X[k, m], db[n] = Σ_m dY[n, m] over one sample chunk.
// Block = one (64·TW)² output tile × one chunk; both operands are (feature, sample) arrays, so a 32-sample slab of
// either is BT contiguous floats per sample and the MFMA operands are read feature-fastest from LDS.
// ------------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dY; const float* X; int H; int M; int chunk; float* pW; float* pB;
  // x3 kernel only: dZ != null ⇒ dY holds h2 and the operand is δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²), formed while staging
  const float* dZ; int ldd; int Ad; const float* W3t;
  // x2 kernel: the head cotangent (bz, ld bld, bA live rows) and wmax bound |δ2| per sample — the block's scale comes from them
  const float* bz; int bld; int bA; const float* wmax;
  // GEN flavour (wide_wgrad_x2_kernel<4, true>): X is not read — h1 is regenerated from the observations (wide_fused.hpp, wide_wgrad_gen_kernel)
  const float* obs = nullptr; const int32_t* perm = nullptr; int D = 0; const float* W1f = nullptr; const float* w1sc = nullptr;
};

template <int TW>
__global__ void __launch_bounds__(256) wide_wgrad_kernel(WgradArgs a) {
  constexpr int BT = 64 * TW;
  __shared__ __attribute__((aligned(16))) float Yl[32 * BT];
  __shared__ __attribute__((aligned(16))) float Xl[32 * BT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int nb = a.H / BT, tnb = blockIdx.y % nb, tkb = blockIdx.y / nb;
  const int n0 = tnb * BT, kk0 = tkb * BT;
  const int wn = wave & 1, wk = wave >> 1;
  f32x16 acc[TW][TW];
#pragma unroll
  for (int x = 0; x < TW; ++x)
#pragma unroll
    for (int y = 0; y < TW; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  f32x4 bacc = {0.0f, 0.0f, 0.0f, 0.0f};
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;
  // thread t stages float4 slots t, t+256, … of the [32][BT] slab: slot → (sample p / BT, rows p % BT); BT·8 slots / 256
  // threads = BT/32 per thread, each a fixed (sample offset, row) pair — pointers are set up once
  constexpr int NU = BT / 32;
  int soff[NU], mmu[NU];   // element offset of the slot inside a slab (rows relative to the block's first row)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int p = 4 * (tid + 256 * u), mm = p / BT, n = p - mm * BT;
    mmu[u] = mm;
    soff[u] = a.H * mm + n;
  }
  const float* ybase = a.dY + (size_t)a.H * c0 + n0;
  const float* xbase = a.X + (size_t)a.H * c0 + kk0;
  const bool do_bias = (tkb == 0) && a.pB;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 yr[NU], xr[NU];
  auto fetch = [&](int m) {
    const int mv = c1 - m;   // samples left in the chunk
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const bool ok = mmu[u] < mv;
      const size_t off = (size_t)a.H * (m - c0);   // wave-uniform
      yr[u] = ok ? *reinterpret_cast<const f32x4*>(ybase + off + soff[u]) : zero4;
      xr[u] = ok ? *reinterpret_cast<const f32x4*>(xbase + off + soff[u]) : zero4;
    }
  };
  if (c0 < c1) fetch(c0);
  for (int m = c0; m < c1; m += 32) {
    if (m != c0) __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      *reinterpret_cast<f32x4*>(Yl + 4 * (tid + 256 * u)) = yr[u];
      *reinterpret_cast<f32x4*>(Xl + 4 * (tid + 256 * u)) = xr[u];
      if (do_bias) bacc += yr[u];
    }
    __syncthreads();
    if (m + 32 < c1) fetch(m + 32);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int mm = 2 * s + hf;
      float av[TW], bv[TW];
#pragma unroll
      for (int x = 0; x < TW; ++x) av[x] = Yl[mm * BT + (wn * TW + x) * 32 + j];
#pragma unroll
      for (int y = 0; y < TW; ++y) bv[y] = Xl[mm * BT + (wk * TW + y) * 32 + j];
#pragma unroll
      for (int x = 0; x < TW; ++x)
#pragma unroll
        for (int y = 0; y < TW; ++y) acc[x][y] = mfma32(av[x], bv[y], acc[x][y]);
    }
  }
  __syncthreads();
  float* pw = a.pW + (size_t)blockIdx.x * a.H * a.H;
#pragma unroll
  for (int x = 0; x < TW; ++x)
#pragma unroll
    for (int y = 0; y < TW; ++y) {
      const int k = kk0 + (wk * TW + y) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * TW + x) * 32 + 8 * g + 4 * hf;
        f32x4 o; o[0] = acc[x][y][4 * g]; o[1] = acc[x][y][4 * g + 1]; o[2] = acc[x][y][4 * g + 2]; o[3] = acc[x][y][4 * g + 3];
        *reinterpret_cast<f32x4*>(pw + (size_t)a.H * k + n) = o;
      }
    }
  if (tkb == 0 && a.pB) {
    // thread t always staged rows (4t mod BT)..+3: fold the 1024/BT threads that share a row quad, in thread order
    *reinterpret_cast<f32x4*>(Yl + 4 * tid) = bacc;
    __syncthreads();
    if (tid < BT) {
      const int quad = tid >> 2, e = tid & 3;
      float s = 0.0f;
      for (int q = 0; q < 1024 / BT; ++q) s += Yl[4 * (quad + (BT / 4) * q) + e];
      a.pB[(size_t)blockIdx.x * a.H + n0 + tid] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe (bf16x3) for 256-wide layers. The reduction runs over SAMPLES, so an
// MFMA operand needs 8 consecutive samples of one row — the strided direction of the (feature, sample) arrays. The
// staging loads are therefore shaped as 4×4 blocks per thread (4 consecutive rows × 4 consecutive samples: four 16-B
// loads), which a thread can transpose in its own registers: each row's 4 samples are split into bf16 hi/mid/lo and
// written with one ds_write_b64 per piece into row-major [row][sample] LDS images (row stride 80 B: conflict-free
// b128 fragment reads). No f32 copy of the slab ever exists in LDS.
// ------------------------------------------------------------------------------------------------------
template <int WNB>   // output tile = (64·WNB rows of dY) × (128 rows of X); 2·WNB waves, each 2×2 MFMA tiles
__global__ void __launch_bounds__(128 * WNB) wide_wgrad_x3_kernel(WgradArgs a) {
  constexpr int BN = 64 * WNB, BK = 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  __bf16* Yp = reinterpret_cast<__bf16*>(smw);                 // [3][BN][X3ROW]
  __bf16* Xp = Yp + 3 * BN * X3ROW;                            // [3][BK][X3ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int nbn = a.H / BN;
  const int tnb = blockIdx.y % nbn, tkb = blockIdx.y / nbn;
  const int n0 = tnb * BN, kk0 = tkb * BK;
  const int wn = wave % WNB, wk = wave / WNB;
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;
  // staging role of this thread: rows 32·wave + 4·ql .. +3, samples 4·sg .. +3 of the slab's 32 (X: the first 4 waves).
  // sg on the low lane bits: the 16 lanes one ds_write_b64 group covers are then 8 sample groups (16 consecutive dwords)
  // of two row quads (80 dwords apart = 16 banks) — conflict-free; with the row quad on the low bits the same store
  // was 4-way conflicted (SQ_LDS_BANK_CONFLICT 21 % of the kernel's CU cycles)
  const int sg = lane & 7, ql = lane >> 3;
  const int rrow = 32 * wave + 4 * ql;
  const bool stage_x = wave < BK / 32;
  const float* ybase = a.dY + (size_t)a.H * c0 + n0 + rrow;
  const float* xbase = a.X + (size_t)a.H * c0 + kk0 + rrow;
  const bool do_bias = (tkb == 0) && a.pB;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 yr[4], xr[4], bacc = zero4;
  // virtual dY: the head's rows for this thread's four hidden units stay in registers for the whole chunk
  const bool virt = a.dZ != nullptr;
  f32x4 w3r[AFUSE];
  if (virt) {
#pragma unroll
    for (int q2 = 0; q2 < AFUSE; ++q2)
      w3r[q2] = q2 < a.Ad ? *reinterpret_cast<const f32x4*>(a.W3t + (size_t)256 * q2 + n0 + rrow) : zero4;
  }
  auto fetch = [&](int m) {
    const size_t off = (size_t)a.H * (m - c0 + 4 * sg);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = m + 4 * sg + e < c1;
      f32x4 y = ok ? *reinterpret_cast<const f32x4*>(ybase + off + (size_t)a.H * e) : zero4;
      if (virt && ok) {
        const float* dz = a.dZ + (size_t)a.ldd * (m + 4 * sg + e);
        f32x4 sacc = zero4;
#pragma unroll
        for (int q2 = 0; q2 < AFUSE; ++q2)
          if (q2 < a.Ad) { const float dv = dz[q2]; sacc += w3r[q2] * dv; }
        y = sacc * (1.0f - y * y);
      }
      yr[e] = y;
      xr[e] = (ok && stage_x) ? *reinterpret_cast<const f32x4*>(xbase + off + (size_t)a.H * e) : zero4;
    }
  };
  if (c0 < c1) fetch(c0);
  for (int m = c0; m < c1; m += 32) {
    if (m != c0) __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {          // row rrow + e: its four samples sit in component e of the four loads
      f32x4 vy;
      vy[0] = yr[0][e]; vy[1] = yr[1][e]; vy[2] = yr[2][e]; vy[3] = yr[3][e];
      uint2 h, mm, l;
      split3x4(vy, h, mm, l);
      *reinterpret_cast<uint2*>(Yp + (0 * BN + rrow + e) * X3ROW + 4 * sg) = h;
      *reinterpret_cast<uint2*>(Yp + (1 * BN + rrow + e) * X3ROW + 4 * sg) = mm;
      *reinterpret_cast<uint2*>(Yp + (2 * BN + rrow + e) * X3ROW + 4 * sg) = l;
      if (stage_x) {
        f32x4 vx;
        vx[0] = xr[0][e]; vx[1] = xr[1][e]; vx[2] = xr[2][e]; vx[3] = xr[3][e];
        split3x4(vx, h, mm, l);
        *reinterpret_cast<uint2*>(Xp + (0 * BK + rrow + e) * X3ROW + 4 * sg) = h;
        *reinterpret_cast<uint2*>(Xp + (1 * BK + rrow + e) * X3ROW + 4 * sg) = mm;
        *reinterpret_cast<uint2*>(Xp + (2 * BK + rrow + e) * X3ROW + 4 * sg) = l;
      }
    }
    if (do_bias) bacc += (yr[0] + yr[1]) + (yr[2] + yr[3]);
    __syncthreads();
    if (m + 32 < c1) fetch(m + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      P3 af[2], bf[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int off = ((wn * 2 + x) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        af[x].hi = *reinterpret_cast<const bf16x8*>(Yp + 0 * BN * X3ROW + off);
        af[x].mid = *reinterpret_cast<const bf16x8*>(Yp + 1 * BN * X3ROW + off);
        af[x].lo = *reinterpret_cast<const bf16x8*>(Yp + 2 * BN * X3ROW + off);
      }
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int off = ((wk * 2 + y) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        bf[y].hi = *reinterpret_cast<const bf16x8*>(Xp + 0 * BK * X3ROW + off);
        bf[y].mid = *reinterpret_cast<const bf16x8*>(Xp + 1 * BK * X3ROW + off);
        bf[y].lo = *reinterpret_cast<const bf16x8*>(Xp + 2 * BK * X3ROW + off);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma_x3(af[x], bf[y], acc[x][y]);
    }
  }
  __syncthreads();
  float* pw = a.pW + (size_t)blockIdx.x * a.H * a.H;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int k = kk0 + (wk * 2 + y) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * 2 + x) * 32 + 8 * g + 4 * hf;
        f32x4 o; o[0] = acc[x][y][4 * g]; o[1] = acc[x][y][4 * g + 1]; o[2] = acc[x][y][4 * g + 2]; o[3] = acc[x][y][4 * g + 3];
        *reinterpret_cast<f32x4*>(pw + (size_t)a.H * k + n) = o;
      }
    }
  if (do_bias) {
    // fold the 8 sample groups of a row quad (lanes 8·ql + sg) in group order; lanes with sg = 0 own rows rrow..rrow+3
    float* scr = reinterpret_cast<float*>(smw);
    *reinterpret_cast<f32x4*>(scr + 4 * tid) = bacc;
    __syncthreads();
    if (sg == 0) {
      f32x4 sacc = bacc;
      for (int q = 1; q < 8; ++q) sacc += *reinterpret_cast<const f32x4*>(scr + 4 * (tid + q));
      *reinterpret_cast<f32x4*>(a.pB + (size_t)blockIdx.x * a.H + n0 + rrow) = sacc;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same weight gradient as fp16x2 (mlp_x2.hpp). The reduction runs over samples, so dY = δ2 needs ONE power-of-two scale
// for everything a block accumulates — and a block owns one sample chunk whose partial is unscaled before it is written, so
// the scale is the block's own: G = 2^(14 − ⌈log2 max_m bound_m⌉) over the chunk's samples, bound_m = Σ_a |δ3[a, m]|·wmax[a]
// (≥ every |δ2[·, m]|; a few KB of reads, no pass over δ2). X = h1 takes the static 2^14.
// ------------------------------------------------------------------------------------------------------
template <int WNB, bool GEN = false>
__global__ void __launch_bounds__(128 * WNB) wide_wgrad_x2_kernel(WgradArgs a) {
  constexpr int BN = 64 * WNB, BK = 128, NT = 128 * WNB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  _Float16* Yp = reinterpret_cast<_Float16*>(smw);             // [2][BN][X3ROW]
  _Float16* Xp = Yp + 2 * BN * X3ROW;                          // [2][BK][X3ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hf = lane >> 5;
  const int nbn = a.H / BN;
  const int tnb = blockIdx.y % nbn, tkb = blockIdx.y / nbn;
  const int n0 = tnb * BN, kk0 = tkb * BK;
  const int wn = wave % WNB, wk = wave / WNB;
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;
  // the block's scale
  float G, Ginv;
  {
    float bmax = 0.0f;
    for (int m = c0 + tid; m < c1; m += NT) {
      float bound = 0.0f;
      for (int q2 = 0; q2 < a.bA; ++q2) bound = __builtin_fmaf(__builtin_fabsf(a.bz[(size_t)a.bld * m + q2]), a.wmax[q2], bound);
      bmax = __builtin_fmaxf(bmax, bound);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bmax = __builtin_fmaxf(bmax, __shfl_xor(bmax, o, 64));
    float* red = reinterpret_cast<float*>(smw);
    if (lane == 0) red[wave] = bmax;
    __syncthreads();
    bmax = red[0];
    for (int w8 = 1; w8 < NT / 64; ++w8) bmax = __builtin_fmaxf(bmax, red[w8]);
    __syncthreads();
    pow2_scale(bmax, G, Ginv);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  const int sg = lane & 7, ql = lane >> 3;
  const int rrow = 32 * wave + 4 * ql;
  const bool stage_x = wave < BK / 32;
  const float* ybase = a.dY + (size_t)a.H * c0 + n0 + rrow;
  const float* xbase = GEN ? nullptr : a.X + (size_t)a.H * c0 + kk0 + rrow;
  const bool do_bias = (tkb == 0) && a.pB;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 yr[4], xr[4], bacc = zero4;
  // GEN: waves 0-3 regenerate their 32-unit tile of the h1 slab with one fp16x2 product (see wide_wgrad_gen_kernel): W1ᵀ fragment + bias in registers
  P2 w1b; float b1u = 0.0f, w1un = 0.0f, xo[8];
  int src_nx = 0;
  if (GEN && stage_x) {
    src_nx = (c0 + j < c1) ? (a.perm ? a.perm[c0 + j] : c0 + j) : 0;
    const f16x8* wf = reinterpret_cast<const f16x8*>(a.W1f) + ((kk0 / 32 + wave) * 2) * 64 + lane;
    w1b.hi = wf[0]; w1b.lo = wf[64];
    b1u = a.W1f[4096 + kk0 + 32 * wave + j]; w1un = a.w1sc[1];
  }
  auto fetch = [&](int m) {
    const size_t off = (size_t)a.H * (m - c0 + 4 * sg);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = m + 4 * sg + e < c1;
      yr[e] = ok ? *reinterpret_cast<const f32x4*>(ybase + off + (size_t)a.H * e) : zero4;
      if (!GEN) xr[e] = (ok && stage_x) ? *reinterpret_cast<const f32x4*>(xbase + off + (size_t)a.H * e) : zero4;
    }
    if (GEN && stage_x) {
      // the observation row index of this slab's sample was loaded a slab earlier (a dependent perm → obs chain inside one fetch stalls the
      // wave for a memory round trip before its MFMAs: 497 instead of ≈300 us per launch)
      const bool ok = m + j < c1;
#pragma unroll
      for (int c = 0; c < 8; ++c) xo[c] = 0.0f;
      load_obs8(a.obs, (size_t)src_nx, a.D, 8 * hf, ok, xo);
      const int mn = m + 32 + j;
      src_nx = mn < c1 ? (a.perm ? a.perm[mn] : mn) : 0;
    }
  };
  if (c0 < c1) fetch(c0);
  for (int m = c0; m < c1; m += 32) {
    if (m != c0) __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x4 vy;
      vy[0] = yr[0][e]; vy[1] = yr[1][e]; vy[2] = yr[2][e]; vy[3] = yr[3][e];
      uint2 hh, ll;
      split2x4(vy, G, hh, ll);
      *reinterpret_cast<uint2*>(Yp + (0 * BN + rrow + e) * X3ROW + 4 * sg) = hh;
      *reinterpret_cast<uint2*>(Yp + (1 * BN + rrow + e) * X3ROW + 4 * sg) = ll;
      if (!GEN && stage_x) {
        f32x4 vx;
        vx[0] = xr[0][e]; vx[1] = xr[1][e]; vx[2] = xr[2][e]; vx[3] = xr[3][e];
        split2x4(vx, X2_ACT_SCALE, hh, ll);
        *reinterpret_cast<uint2*>(Xp + (0 * BK + rrow + e) * X3ROW + 4 * sg) = hh;
        *reinterpret_cast<uint2*>(Xp + (1 * BK + rrow + e) * X3ROW + 4 * sg) = ll;
      }
    }
    if (GEN && stage_x) {
      float mx = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c) mx = __builtin_fmaxf(mx, __builtin_fabsf(xo[c]));
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
      float sx, ix;
      pow2_scale(mx, sx, ix);
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = xo[c] * sx;
      const P2 xa = split2(v);
      f32x16 c16;
#pragma unroll
      for (int r = 0; r < 16; ++r) c16[r] = 0.0f;
      c16 = mfma_x2(xa, w1b, c16);                              // rows = samples (registers), columns = units (lanes)
      const float un1 = ix * w1un;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = tanh_exp2_arg(__builtin_fmaf(c16[4 * g + e], un1, b1u), X2_ACT_SCALE);
        uint2 hh, ll;
        split2x4(hv, 1.0f, hh, ll);
        *reinterpret_cast<uint2*>(Xp + (0 * BK + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = hh;
        *reinterpret_cast<uint2*>(Xp + (1 * BK + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = ll;
      }
    }
    if (do_bias) bacc += (yr[0] + yr[1]) + (yr[2] + yr[3]);
    __syncthreads();
    if (m + 32 < c1) fetch(m + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      P2 af[2], bf[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int off = ((wn * 2 + x) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        af[x].hi = *reinterpret_cast<const f16x8*>(Yp + 0 * BN * X3ROW + off);
        af[x].lo = *reinterpret_cast<const f16x8*>(Yp + 1 * BN * X3ROW + off);
      }
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int off = ((wk * 2 + y) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        bf[y].hi = *reinterpret_cast<const f16x8*>(Xp + 0 * BK * X3ROW + off);
        bf[y].lo = *reinterpret_cast<const f16x8*>(Xp + 1 * BK * X3ROW + off);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma_x2(af[x], bf[y], acc[x][y]);
    }
  }
  __syncthreads();
  const float un = Ginv * (1.0f / X2_ACT_SCALE);
  float* pw = a.pW + (size_t)blockIdx.x * a.H * a.H;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int k = kk0 + (wk * 2 + y) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * 2 + x) * 32 + 8 * g + 4 * hf;
        f32x4 o; o[0] = acc[x][y][4 * g] * un; o[1] = acc[x][y][4 * g + 1] * un; o[2] = acc[x][y][4 * g + 2] * un; o[3] = acc[x][y][4 * g + 3] * un;
        *reinterpret_cast<f32x4*>(pw + (size_t)a.H * k + n) = o;
      }
    }
  if (do_bias) {
    float* scr = reinterpret_cast<float*>(smw);
    *reinterpret_cast<f32x4*>(scr + 4 * tid) = bacc;
    __syncthreads();
    if (sg == 0) {
      f32x4 sacc = bacc;
      for (int q = 1; q < 8; ++q) sacc += *reinterpret_cast<const f32x4*>(scr + 4 * (tid + q));
      *reinterpret_cast<f32x4*>(a.pB + (size_t)blockIdx.x * a.H + n0 + rrow) = sacc;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Skinny weight gradients on the VALU: out[row, s] = Σ_m Big[row, m]·Small[s, m] (+ Σ_m Big[row, m]) with S ≤ 16.
//   dW1 = dH1·Xᵀ  (Big = dH1, Small = the gathered observations, bias sum = db1)
//   dW3ᵀ = H2·dZᵀ (Big = H2, Small = the head cotangent)
// A thread owns one hidden row; the Small values of a sample are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
  const float* Big; int H; const float* Small; int lds; const int32_t* idx; int M; int chunk;
  float* pW; int os_row, os_s, St, wsize; float* pB;
  // D2 variant (dW3 pass only): the same sweep over h2 also emits δ2[k, m] = (Σ_a W3[a, k]·δ3[a, m])·(1 − h2[k, m]²) — the
  // hidden-layer cotangent both 256-wide backward GEMMs read — so h2 is read once for the two and the K ≤ 8 MFMA pass is gone
  float* D2out; const float* W3t;
};

template <int S, bool UNI, bool D2 = false>
__global__ void __launch_bounds__(256) wide_skinny_kernel(SkinnyArgs a) {
  // a thread owns 4 consecutive hidden rows (one 16-B load per sample); R4 = H/4 threads cover a sample and the
  // block's 256/R4 groups take samples round-robin. UNI (H = 256): a group is a whole wave, Small goes through SGPRs.
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  const int tid = threadIdx.x;
  const int R4 = a.H >> 2, G = 256 / R4;
  const int r4 = tid % R4;
  int g = tid / R4;
  if (UNI) g = __builtin_amdgcn_readfirstlane(g);
  f32x4 acc[S], bacc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = bacc;
  f32x4 w3r[D2 ? S : 1];
  if (D2) {
#pragma unroll
    for (int s = 0; s < S; ++s) w3r[s] = s < a.St ? *reinterpret_cast<const f32x4*>(a.W3t + (size_t)s * a.H + 4 * r4) : bacc;
  }
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;
  for (int m = c0 + g; m < c1; m += 4 * G) {
    f32x4 big[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mu = m + u * G;
      big[u] = bacc * 0.0f;
      if (mu < c1) big[u] = *reinterpret_cast<const f32x4*>(a.Big + (size_t)a.H * mu + 4 * r4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int mu = m + u * G;
      if (mu < c1) {
        int sr = a.idx ? a.idx[mu] : mu;
        if (UNI) sr = __builtin_amdgcn_readfirstlane(sr);
        const float* sp = a.Small + (size_t)sr * a.lds;
        f32x4 t = bacc * 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float sv = s < a.St ? sp[s] : 0.0f;
          acc[s] += big[u] * sv;
          if (D2) t += w3r[s] * sv;
        }
        if (D2) *reinterpret_cast<f32x4*>(a.D2out + (size_t)a.H * mu + 4 * r4) = t * (1.0f - big[u] * big[u]);
        bacc += big[u];
      }
    }
  }
  // fold the G groups in group order, one float4 column at a time
#pragma unroll
  for (int s = 0; s <= S; ++s) {
    if (s == S && !a.pB) break;
    f32x4 v = s < S ? acc[s < S ? s : 0] : bacc;
    if (G > 1) {
      __syncthreads();
      *reinterpret_cast<f32x4*>(red + 4 * tid) = v;
      __syncthreads();
      if (g == 0)
        for (int q = 1; q < G; ++q) v += *reinterpret_cast<const f32x4*>(red + 4 * (tid + q * R4));
    }
    if (g == 0) {
      if (s < S) {
        if (s < a.St) {
          float* pw = a.pW + (size_t)blockIdx.x * a.wsize + (size_t)s * a.os_s;
#pragma unroll
          for (int e = 0; e < 4; ++e) pw[(4 * r4 + e) * a.os_row] = v[e];
        }
      } else {
        *reinterpret_cast<f32x4*>(a.pB + (size_t)blockIdx.x * a.H + 4 * r4) = v;
      }
    }
  }
}

template <int S>
static void skinny_go(hipStream_t st, int blocks, const SkinnyArgs& a) {
  if (a.D2out) {
    if (a.H == 256) hipLaunchKernelGGL((wide_skinny_kernel<S, true, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wide_skinny_kernel<S, false, true>), dim3(blocks), dim3(256), 0, st, a);
    return;
  }
  if (a.H == 256) hipLaunchKernelGGL((wide_skinny_kernel<S, true>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wide_skinny_kernel<S, false>), dim3(blocks), dim3(256), 0, st, a);
}

static int skinny_launch(hipStream_t st, int blocks, const SkinnyArgs& a) {
  if (a.St <= 4) skinny_go<4>(st, blocks, a);
  else if (a.St <= 8) skinny_go<8>(st, blocks, a);
  else if (a.St <= 16) skinny_go<16>(st, blocks, a);
  else {
    // obs_dim up to 64: sixteen columns per launch
    for (int s0 = 0; s0 < a.St; s0 += 16) {
      SkinnyArgs b = a;
      b.Small = a.Small + s0; b.St = (a.St - s0) < 16 ? (a.St - s0) : 16; b.pW = a.pW + (size_t)s0 * a.os_s;
      if (s0 > 0) b.pB = nullptr;
      skinny_go<16>(st, blocks, b);
    }
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Per-sample pieces shared by the act / logprob / loss kernels (runtime n_act ≤ 16, same operation order as
// softmax_logsoftmax<A> and sample_weights<A> in common.hpp)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void softmax_rt(const float (&z)[AMAX], int A, float (&p)[AMAX], float (&lp)[AMAX]) {
  float m = z[0];
#pragma unroll
  for (int a = 1; a < AMAX; ++a) if (a < A) m = fmaxf(m, z[a]);
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) { p[a] = expf(z[a] - m); s += p[a]; }
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) p[a] = p[a] / s;
  float ls = 0.0f;
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) { lp[a] = z[a] - m; ls += expf(lp[a]); }
  const float l = logf(ls);
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) lp[a] = lp[a] - l;
}
__device__ __forceinline__ int sample_rt(const float (&p)[AMAX], int A, double u) {
  float sw = 0.0f;
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) sw += p[a];
  const double t = u * (double)sw;
  int i = 0;
  float cw = p[0];
#pragma unroll
  for (int a = 1; a < AMAX; ++a) {
    const bool go = (a < A) && ((double)cw < t) && (i == a - 1);
    i = go ? a : i;
    cw = go ? cw + p[a] : cw;
  }
  return i;
}
__device__ __forceinline__ float pick_rt(const float (&v)[AMAX], int A, int i) {
  float r = v[0];
#pragma unroll
  for (int a = 1; a < AMAX; ++a) if (a < A) r = (i == a) ? v[a] : r;
  return r;
}
__device__ __forceinline__ void load_logits(const float* Z, int A8, int A, size_t m, float (&z)[AMAX]) {
#pragma unroll
  for (int a = 0; a < AMAX; ++a) z[a] = a < A ? Z[(size_t)A8 * m + a] : 0.0f;
}

// get_action on precomputed logits (ppo.jl:23-31) with caller-supplied uniforms
__global__ void __launch_bounds__(256) wide_sample_kernel(const float* __restrict__ Z, int A8, int A, const float* __restrict__ V,
                                                         const double* __restrict__ u, int n, int32_t* __restrict__ action,
                                                         float* __restrict__ logprob, float* __restrict__ value) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  float z[AMAX], p[AMAX], lp[AMAX];
  load_logits(Z, A8, A, (size_t)b, z);
  softmax_rt(z, A, p, lp);
  const int act = sample_rt(p, A, u[b]);
  action[b] = act;
  logprob[b] = pick_rt(lp, A, act);
  if (value) value[b] = V[b];
}

// logprob_actions on precomputed logits (ppo.jl:36-44)
__global__ void __launch_bounds__(256) wide_logprob_kernel(const float* __restrict__ Z, int A8, int A, const int32_t* __restrict__ actions,
                                                          int n, float* __restrict__ logprob, float* __restrict__ entropy) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n) return;
  float z[AMAX], p[AMAX], lp[AMAX];
  load_logits(Z, A8, A, (size_t)b, z);
  softmax_rt(z, A, p, lp);
  logprob[b] = pick_rt(lp, A, actions[b]);
#pragma unroll
  for (int a = 0; a < AMAX; ++a) if (a < A) entropy[(size_t)A * b + a] = -(p[a] * lp[a]);
}

// ------------------------------------------------------------------------------------------------------
// One rollout step after the two forward passes: sampling, env step, Buffer.add!, episode bookkeeping
// (ppo.jl:125-165). One thread per env.
// ------------------------------------------------------------------------------------------------------
struct WStepArgs {
  DevCfg c; const float* Z; int A8; const float* V;
  float* obs; int32_t* action; float* logprob; float* reward; uint8_t* terminal; float* value;
  float* env_state; int32_t* env_t; float* cur_obs; uint8_t* next_done; float* ep_return; int32_t* ep_length; double* ep_stats;
  crl_episode_record* ring; uint32_t* ring_count; int ring_cap;
  uint64_t iteration; int step;
};

// one env, one step; the episode statistics of a finished episode are added to the caller's running sums. zreg: the env's logits when the caller holds them
// in registers (else they come from a.Z); xout: receives the env's next observation (16 floats, zero beyond obs_dim) besides cur_obs; a.V may be null (the
// caller fills the value buffer later: wide_rs_rollout_kernel)
__device__ __forceinline__ void wide_step_env(const WStepArgs& a, int e, int step, double& st_n, double& st_ret, double& st_len, double& st_max,
                                              const float* zreg = nullptr, float* xout = nullptr) {
  const DevCfg& c = a.c;
  const int D = c.D, A = c.A;
  const uint32_t gid = c.env_id_offset + (uint32_t)e;
  const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
  const size_t b = (size_t)e + (size_t)c.nt * step;
  int ep_len = a.ep_length[e] + 1;                                   // ppo.jl:125
  float z[AMAX], p[AMAX], lp[AMAX];
  if (zreg) {
#pragma unroll
    for (int i = 0; i < AMAX; ++i) z[i] = (i < A && i < 8) ? zreg[i & 7] : 0.0f;
  } else load_logits(a.Z, a.A8, A, (size_t)e, z);
  softmax_rt(z, A, p, lp);                                          // ppo.jl:127 get_action
  const double u = u53(philox_env(c.seed, gid, gstep, 0));
  const int act = sample_rt(p, A, u);
  const float lpa = pick_rt(lp, A, act);
  float* co = a.cur_obs + (size_t)D * e;
  float* es = a.env_state + (size_t)D * e;
  float* ob = a.obs + b * (size_t)D;
  for (int i = 0; i < D; ++i) ob[i] = co[i];                        // ppo.jl:133-140 Buffer.add!
  a.action[b] = act; a.logprob[b] = lpa; a.terminal[b] = a.next_done[e]; if (a.V) a.value[b] = a.V[e];
  bool done; float rew;
  if (c.env_kind == CRL_ENV_CARTPOLE) {
    float s[4] = {es[0], es[1], es[2], es[3]};
    int t_env = a.env_t[e];
    done = cartpole_step(s, t_env, act);                             // ppo.jl:130
    rew = done ? 0.0f : 1.0f;                                        // ppo.jl:132
    float so[4] = {s[0], s[1], s[2], s[3]};                          // ppo.jl:143: the observation is taken before the reset (Q7)
    if (done) {
      cartpole_reset(s, c.seed, gid, gstep, 1);                      // ppo.jl:164
      t_env = 0;
      if (!c.stale_obs) for (int i = 0; i < 4; ++i) so[i] = s[i];
    }
    for (int i = 0; i < 4; ++i) { co[i] = so[i]; es[i] = s[i]; }
    a.env_t[e] = t_env;
    if (xout) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xout[i] = i < 4 ? so[i] : 0.0f;
    }
  } else {
    if (xout) {
#pragma unroll
      for (int i = 0; i < 16; ++i) xout[i] = 0.0f;
    }
    for (int q = 0; 4 * q < D; ++q) {
      float o4[4];
      synth_obs4(c.seed, gid, gstep, q, o4);
      for (int i = 0; i < 4 && 4 * q + i < D; ++i) { es[4 * q + i] = o4[i]; co[4 * q + i] = o4[i]; }
      if (xout) {                                                    // constant indices: the caller's array stays in registers
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (q == qq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xout[4 * qq + i] = 4 * qq + i < D ? o4[i] : 0.0f;
          }
      }
    }
    synth_reward_done(c.seed, gid, gstep, rew, done);
  }
  a.reward[b] = rew;
  a.next_done[e] = done ? 1 : 0;                                     // ppo.jl:144
  float ep_ret = a.ep_return[e] + rew;                               // ppo.jl:145
  if (done) {                                                        // ppo.jl:147-165
    st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, fmax(0.0, (double)ep_ret));
    if (a.ring_cap > 0) {
      const uint32_t slot = atomicAdd(a.ring_count, 1u);
      if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
    }
    ep_ret = 0.0f; ep_len = 0;
  }
  a.ep_return[e] = ep_ret; a.ep_length[e] = ep_len;
}
// a wave's episode statistics into the handle's four accumulators
__device__ __forceinline__ void wide_step_stats(double* ep_stats, double st_n, double st_ret, double st_len, double st_max) {
  st_n = wave_sum(st_n);
  if (st_n > 0.0) {
    st_ret = wave_sum(st_ret); st_len = wave_sum(st_len);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) st_max = fmax(st_max, __shfl_xor(st_max, o, 64));
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&ep_stats[0], st_n); atomicAdd(&ep_stats[1], st_ret); atomicAdd(&ep_stats[2], st_len);
      atomicMax(reinterpret_cast<unsigned long long*>(&ep_stats[3]), (unsigned long long)__double_as_longlong(st_max));
    }
  }
}

__global__ void __launch_bounds__(256) wide_step_kernel(WStepArgs a) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  if (e < a.c.nt) wide_step_env(a, e, a.step, st_n, st_ret, st_len, st_max);
  wide_step_stats(a.ep_stats, st_n, st_ret, st_len, st_max);
}

// ------------------------------------------------------------------------------------------------------
// The whole rollout of the 2×256 fp16x2 configuration as ONE launch (ppo.jl:123-166). Envs are independent, so a block owns a tile
// of 32 envs for all num_steps steps: per step layer 1 of both networks on the VALU (thread = hidden row, its 8 + 1 weights stay in
// registers for the launch; the actor with tanh_fast, the critic with the exp2 activation as in the per-step path), then the two
// 256×256 layers with their fused heads by the per-step path's own block body (wide_dense_x2_body<EPI_TANH, 1>: W2 streams from L2,
// h1 passes through a 32 KB per-block slice of the workspace that never leaves the caches), then 32 threads sample, step the env
// and append to the buffer. Was three launches per step (57 µs x 128 steps at C3).
// ------------------------------------------------------------------------------------------------------
// The layer body inlined into the step loop: everything it derives from (thread index, argument) pairs is loop-invariant, gets hoisted
// out of the step loop for both networks, and the kernel then needs 227 registers (one block per CU) or spills 448 bytes at 128. The
// arguments those values hang on are made opaque once per step, which keeps the per-step body what it is in the per-step kernel.
__device__ __forceinline__ void wide_rollout_layer2(DenseX3Args x) {
  asm volatile("" : "+s"(x.Wx3), "+s"(x.X), "+s"(x.Y), "+s"(x.Z), "+s"(x.bias), "+s"(x.W3t), "+s"(x.M));
  wide_dense_x2_body<EPI_TANH, 1>(x);
}
struct WRollArgs {
  DenseX3Args x[2]; WStepArgs s;
  const float* W1[2]; const float* b1[2]; float* h1[2]; int fast[2]; int D; size_t obs_off;   // obs_off: byte offset of the obs tile in LDS
};
__global__ void __launch_bounds__(512, 4) wide_rollout_persist_kernel(WRollArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  float* xt = reinterpret_cast<float*>(smx + r.obs_off);                 // [32 envs][D] current observations of the tile
  const int tid = threadIdx.x, net = tid >> 8, n = tid & 255, D = r.D;
  const int m0 = blockIdx.x * 32, nt = r.s.c.nt;
  float w1[16];                                                          // this thread's row of W1 (obs_dim <= 16) and its bias
#pragma unroll
  for (int k = 0; k < 16; ++k) w1[k] = k < D ? r.W1[net][n + 256 * k] : 0.0f;
  const float bias1 = r.b1[net][n];
  const bool fast = r.fast[net] != 0;
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  for (int step = 0; step < r.s.c.k; ++step) {
    for (int i = tid; i < 32 * D; i += 512) { const int m = m0 + i / D; xt[i] = m < nt ? r.s.cur_obs[(size_t)m0 * D + i] : 0.0f; }
    __syncthreads();
    {
      float* h1 = r.h1[net] + (size_t)m0 * 256 + n;
#pragma unroll 2
      for (int m = 0; m < 32; ++m) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < D) acc = __builtin_fmaf(w1[k], xt[m * D + k], acc);
        if (m0 + m < nt) h1[(size_t)m * 256] = wide_tanh(acc + bias1, fast);
      }
    }
    __syncthreads();
    wide_rollout_layer2(r.x[0]);
    __syncthreads();
    wide_rollout_layer2(r.x[1]);
    __syncthreads();
    int e = m0 + tid;
    asm volatile("" : "+v"(e));          // as above: the dozen per-env addresses are formed per step, not kept across the loop
    if (tid < 32 && e < nt) wide_step_env(r.s, e, step, st_n, st_ret, st_len, st_max);
    __syncthreads();
  }
  if (tid < 64) wide_step_stats(r.s.ep_stats, st_n, st_ret, st_len, st_max);
}

// ------------------------------------------------------------------------------------------------------
// The same rollout in the producer / consumer form of the fused forward pass (wide_fused.hpp), option wide_rollout_persist = 2 (default
// when num_envs is a multiple of 64). wide_rollout_persist_kernel walks 16 weight slabs per step through registers with two barriers
// each and a fetch distance of one short MFMA phase: 52 µs per step at C3, almost all of it L2 latency. Here a block owns 64 envs and
//   waves 0-3  multiply: rows 64c … 64c + 63 x 64 envs (2 x 2 accumulator tiles), then tanh + head partials; wave 0 also steps the envs;
//   waves 4-7  bring the next 32 KB weight slab by LDS-DMA (8 pieces each) and make the next h1 slab (layer 1 as ONE fp16x2 product per
//              32 x 32 tile — computed by both waves of a pair, each of which then finishes half of it: tanh, split, LDS stores);
// one barrier per slab, actor then critic, the critic's first slab prepared under the actor's epilogue. Nothing but the logits / values
// (for wide_step_env) and the rollout buffer leaves the CU. The actor keeps the reference's rational tanh_fast in both layers; its layer 1
// is now an fp16x2 product like its layer 2 (action indices equal the oracle's away from CDF knots — the margin rule of the parity tests).
// ------------------------------------------------------------------------------------------------------
constexpr int RP_MB = 64;
constexpr int RP_XBYTES = 2 * RP_MB * X3ROW * 2;             // 10,240: one activation slab, [piece][env][X3ROW halves]
constexpr int RP_OFF_X = 2 * FX_WBYTES;                       // 65,536
constexpr int RP_W1F_BYTES = 16384 + 1024;                    // per network: W1 fragments + bias table (pack: w1f)
constexpr int RP_OFF_W1F = RP_OFF_X + 2 * RP_XBYTES;          // 86,016
constexpr int RP_OFF_W3 = RP_OFF_W1F + 2 * RP_W1F_BYTES;      // 120,832: actor W3ᵀ [A <= 8][256] f32, then the critic's [256]
constexpr int RP_OFF_B2 = RP_OFF_W3 + 9 * 1024;               // b2 of the actor, of the critic
constexpr int RP_OFF_HP = RP_OFF_B2 + 2 * 1024;               // head partials [4 row groups][64 envs][8] f32
constexpr int RP_LDS = RP_OFF_HP + 4 * RP_MB * 8 * 4;         // 140,288 bytes
constexpr float INV_TWO_LOG2E = 0.34657359027997264f;
struct RollPCNet {
  const float* W1f; const float* w1sc; const float* Wx2; const float* b2; const float* wsc; const float* W3t; const float* b3;
  float* Z; int A; int ldz; int rat;                          // rat: 1 = tanh_fast (rational), 0 = the exp2 form
};
struct RollPCArgs { RollPCNet n[2]; WStepArgs s; int D; };

template <int DP>
__global__ void __launch_bounds__(512) wide_rollout_pc_kernel(RollPCArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const int m0 = blockIdx.x * RP_MB;
  for (int net = 0; net < 2; ++net) {
    for (int i = tid; i < RP_W1F_BYTES / 16; i += 512)
      reinterpret_cast<f32x4*>(smx + RP_OFF_W1F + net * RP_W1F_BYTES)[i] = reinterpret_cast<const f32x4*>(r.n[net].W1f)[i];
    for (int i = tid; i < r.n[net].A * 256; i += 512) reinterpret_cast<float*>(smx + RP_OFF_W3 + net * 8192)[i] = r.n[net].W3t[i];
    if (tid < 256) reinterpret_cast<float*>(smx + RP_OFF_B2 + net * 1024)[tid] = r.n[net].b2[tid];
  }
  __syncthreads();
  const int nsteps = r.s.c.k;
  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ producers
    const int p = wave - 4;
    f16x8 xhi, xlo; float xi1 = 0.0f;
    // every producer: 8 of the slab's 32 weight pieces, and HALF of an h1 tile (envs 32·(p & 1) …, registers 8·(p >> 1) … of the 32 x 32
    // product, which both waves of a pair compute — three MFMAs are cheaper than the 8 rational tanh they spare each wave)
    const int pt = p & 1, ph = p >> 1;
    // layer 1 of a network for the whole step (8 slabs) in one burst while the matrix pipe is idle — the env step for the actor, the actor's
    // epilogue for the critic — as in wide_fused_fwd_pc_kernel
    f32x16 hpre[8];
    auto layer1 = [&](int net) {
      const unsigned char* tab = smx + RP_OFF_W1F + net * RP_W1F_BYTES;
      P2 bf; bf.hi = xhi; bf.lo = xlo;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f16x8* wf = reinterpret_cast<const f16x8*>(tab) + (s * 2) * 64 + lane;
        P2 af; af.hi = wf[0]; af.lo = wf[64];
#pragma unroll
        for (int q = 0; q < 16; ++q) hpre[s][q] = 0.0f;
        hpre[s] = mfma_x2(af, bf, hpre[s]);
      }
    };
    auto produce = [&](int net, int s, const f32x16& c, unsigned char* wbuf, unsigned char* xbuf, bool with_dma) {
      if (with_dma) {      // a network's FIRST weight slab only; inside the slab loop the consumers fetch the next one behind their own MFMAs
        const char* g = reinterpret_cast<const char*>(r.n[net].Wx2) + (size_t)s * FX_WBYTES + p * 8192;
        const unsigned lds0 = lds_addr_of(wbuf) + p * 8192, voff = lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_dma16(g + i * 1024, voff, lds0 + i * 1024);
      }
      {
        const unsigned char* tab = smx + RP_OFF_W1F + net * RP_W1F_BYTES;
        const float xinv = xi1 * r.n[net].w1sc[1];
        const float* b1l = reinterpret_cast<const float*>(tab + 16384) + 32 * s + 4 * hf;
        _Float16* Xl = reinterpret_cast<_Float16*>(xbuf);
        const bool rat = r.n[net].rat != 0;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {                                 // registers 4q4 … 4q4 + 3 = units 32s + 8q4 + 4hf + {0..3}
          const int q4 = 2 * ph + qq;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b1l + 8 * q4);
          f32x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float cv = ph ? c[8 + 4 * qq + e] : c[4 * qq + e];
            const float t = __builtin_fmaf(cv, xinv, bv[e]);               // 2·log2(e)·(W1·x + b1)
            hv[e] = rat ? tanh_fast(t * INV_TWO_LOG2E) * X2_ACT_SCALE : tanh_exp2_arg(t, X2_ACT_SCALE);
          }
          uint2 hh, ll;
          split2x4(hv, 1.0f, hh, ll);
          *reinterpret_cast<uint2*>(Xl + (32 * pt + j) * X3ROW + 8 * q4 + 4 * hf) = hh;
          *reinterpret_cast<uint2*>(Xl + RP_MB * X3ROW + (32 * pt + j) * X3ROW + 8 * q4 + 4 * hf) = ll;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    };
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      __builtin_amdgcn_s_barrier();                                      // B_obs: the envs have been stepped
      {
        const int gm = m0 + 32 * pt + j;
        float xr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int cc = 8 * hf + q; xr[q] = (cc < r.D && cc < DP) ? r.s.cur_obs[(size_t)gm * r.D + cc] : 0.0f; }
        float m = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) m = __builtin_fmaxf(m, __builtin_fabsf(xr[q]));
        m = __builtin_fmaxf(m, xor32(m));
        float s1;
        pow2_scale(m, s1, xi1);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = xr[q] * s1;
        const P2 q2 = split2(v);
        xhi = q2.hi; xlo = q2.lo;
      }
      layer1(0);
      produce(0, 0, hpre[0], smx, smx + RP_OFF_X, true);
#pragma unroll 1
      for (int net = 0; net < 2; ++net) {
        __builtin_amdgcn_s_barrier();                                    // B_start
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s < 7) produce(net, s + 1, hpre[(s + 1) & 7], smx + ((s + 1) & 1) * FX_WBYTES, smx + RP_OFF_X + ((s + 1) & 1) * RP_XBYTES, false);
          __builtin_amdgcn_s_barrier();
        }
        if (net == 0) { layer1(1); produce(1, 0, hpre[0], smx, smx + RP_OFF_X, true); }   // the critic's layer 1 and first slab, under the actor's epilogue
        __builtin_amdgcn_s_barrier();                                    // B_epi
      }
      __builtin_amdgcn_s_barrier();                                      // B_fold
    }
  } else {
    // ------------------------------------------------------------------------------------------------ consumer c: rows 64c … 64c + 63
    const int c = wave;
    double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
    float* hp_all = reinterpret_cast<float*>(smx + RP_OFF_HP);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
      __builtin_amdgcn_s_barrier();                                      // B_obs
#pragma unroll 1
      for (int net = 0; net < 2; ++net) {
        const RollPCNet& nn = r.n[net];
        f32x16 acc[2][2];
#pragma unroll
        for (int ai = 0; ai < 2; ++ai)
#pragma unroll
          for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ai][bi][q] = 0.0f;
        __builtin_amdgcn_s_barrier();                                    // B_start
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
          const f16x8* Wl = reinterpret_cast<const f16x8*>(smx + (s & 1) * FX_WBYTES);
          const _Float16* Xl = reinterpret_cast<const _Float16*>(smx + RP_OFF_X + (s & 1) * RP_XBYTES);
          // the next weight slab: this consumer's 8 of the 32 pieces, one behind every product (as in wide_fused_fwd_pc_kernel)
          const bool dma = s + 1 < 8;
          const char* wg = reinterpret_cast<const char*>(nn.Wx2) + (size_t)(s + 1) * FX_WBYTES + c * 8192;
          const unsigned wl0 = lds_addr_of(smx + ((s + 1) & 1) * FX_WBYTES) + c * 8192, wvo = lane * 16;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            P2 af[2], bf[2];
#pragma unroll
            for (int ai = 0; ai < 2; ++ai) {
              const int fr = (ks * 8 + 2 * c + ai) * 64 + lane;
              af[ai].hi = Wl[fr]; af[ai].lo = Wl[1024 + fr];
            }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
              const int off = (32 * bi + j) * X3ROW + 16 * ks + 8 * hf;
              bf[bi].hi = *reinterpret_cast<const f16x8*>(Xl + off);
              bf[bi].lo = *reinterpret_cast<const f16x8*>(Xl + RP_MB * X3ROW + off);
            }
#pragma unroll
            for (int ai = 0; ai < 2; ++ai)
#pragma unroll
              for (int bi = 0; bi < 2; ++bi) {
                acc[ai][bi] = mfma_x2(af[ai], bf[bi], acc[ai][bi]);
                if (dma) { const int pc = ks * 4 + ai * 2 + bi; lds_dma16(wg + pc * 1024, wvo, wl0 + pc * 1024); }
              }
          }
          if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        // epilogue: h2 = tanh(acc·unscale + b2) stays in registers; head partials of this row group
        const float cs = nn.wsc[1] * (1.0f / X2_ACT_SCALE);
        const bool rat = nn.rat != 0;
        const int hs = nn.A;
        float hacc[2][PC_AMAX];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
          for (int aa = 0; aa < PC_AMAX; ++aa) hacc[bi][aa] = 0.0f;
#pragma unroll
        for (int ai = 0; ai < 2; ++ai) {
          const int n0 = 64 * c + 32 * ai;
          const float* b2l = reinterpret_cast<const float*>(smx + RP_OFF_B2 + net * 1024) + n0 + 4 * hf;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b2l + 8 * g);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = __builtin_fmaf(acc[ai][bi][4 * g + e], cs, bv[e]);
                acc[ai][bi][4 * g + e] = rat ? tanh_fast(x) : tanh_exp2(x, TWO_LOG2E, 1.0f);
              }
          }
#pragma unroll
          for (int aa = 0; aa < PC_AMAX; ++aa) {
            if (aa < nn.A) {
              const float* w3l = reinterpret_cast<const float*>(smx + RP_OFF_W3 + net * 8192) + 256 * aa + n0 + 4 * hf;
              f32x4 w[4];
#pragma unroll
              for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const f32x4*>(w3l + 8 * g);
#pragma unroll
              for (int bi = 0; bi < 2; ++bi) {
                float pp = hacc[bi][aa];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                  for (int e = 0; e < 4; ++e) pp = __builtin_fmaf(w[g][e], acc[ai][bi][4 * g + e], pp);
                hacc[bi][aa] = pp;
              }
            }
          }
        }
#pragma unroll
        for (int aa = 0; aa < PC_AMAX; ++aa) {
          if (aa < nn.A) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
              float pp = hacc[bi][aa];
              pp += xor32(pp);
              if (hf == 0) hp_all[c * (RP_MB * hs) + (32 * bi + j) * hs + aa] = pp;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // B_epi
        for (int i = tid; i < RP_MB * nn.A; i += 256) {                  // the four row groups' partials, fixed order
          const int m = i / nn.A, aa = i - m * nn.A;
          float z = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) z += hp_all[q * (RP_MB * hs) + m * hs + aa];
          nn.Z[(size_t)nn.ldz * (m0 + m) + aa] = z + nn.b3[aa];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                      // B_fold: logits and values of the 64 envs are out
      if (wave == 0) {
        wide_step_env(r.s, m0 + lane, step, st_n, st_ret, st_len, st_max);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (wave == 0) wide_step_stats(r.s.ep_stats, st_n, st_ret, st_len, st_max);
  }
}

}  // namespace crl
#include "wide_rs.hpp"
namespace crl {

// env construction for the synthetic env (oracle: orc_env_init, gstep = ~0)
__global__ void __launch_bounds__(256) wide_synth_reset_kernel(DevCfg c, float* env_state, int32_t* env_t, float* cur_obs,
                                                              uint8_t* next_done, float* ep_return, int32_t* ep_length, double* ep_stats) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e == 0) for (int i = 0; i < 4; ++i) ep_stats[i] = 0.0;
  if (e >= c.nt) return;
  const uint32_t gid = c.env_id_offset + (uint32_t)e;
  for (int q = 0; 4 * q < c.D; ++q) {
    float o4[4];
    synth_obs4(c.seed, gid, ~(uint64_t)0, q, o4);
    for (int i = 0; i < 4 && 4 * q + i < c.D; ++i) { env_state[(size_t)c.D * e + 4 * q + i] = o4[i]; cur_obs[(size_t)c.D * e + 4 * q + i] = o4[i]; }
  }
  env_t[e] = 0; next_done[e] = 0; ep_return[e] = 0.0f; ep_length[e] = 0;
}

// ------------------------------------------------------------------------------------------------------
// Value-loss scalar u = mean(newvalue .- mb_returns .^ 2) (ppo.jl:232, Q4) and #{b : u > q_b}
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wide_vsum_kernel(const float* __restrict__ V, const float* __restrict__ returns,
                                                       const int32_t* __restrict__ perm, int M, double* __restrict__ vpart) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int pos = blockIdx.x * 256 + threadIdx.x; pos < M; pos += gridDim.x * 256) {
    const float R = returns[perm[pos]];
    s += (double)(V[pos] - R * R);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) vpart[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// vfix[5] = Σ(v − R²) of this shard (all-reduced under DP); vfix[6] (as u64) = 0
__global__ void __launch_bounds__(64) wide_vsum_final_kernel(const double* __restrict__ vpart, int n, double* vfix) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += vpart[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) { vfix[5] = s; *reinterpret_cast<unsigned long long*>(&vfix[6]) = 0ull; }
}
// vpart != nullptr (one GPU): every block folds the nparts partial sums of wide_vsum_kernel itself — in wide_vsum_final_kernel's order, so all blocks and
// the two-launch form get the same bits — and block 0 leaves Σ, u and, when u <= 0 (always, in practice: q_b >= 0), the count 0: two launches fewer per step
__global__ void __launch_bounds__(256) wide_vcount_kernel(DevCfg c, const float* __restrict__ V, const float* __restrict__ values,
                                                         const float* __restrict__ returns, const int32_t* __restrict__ perm, int M,
                                                         double Mglobal, double* vfix, const double* __restrict__ vpart, int nparts) {
  __shared__ double ssum;
  double total;
  if (vpart) {
    if (threadIdx.x < 64) {
      double s = 0.0;
      for (int i = threadIdx.x; i < nparts; i += 64) s += vpart[i];
      s = wave_sum(s);
      if (threadIdx.x == 0) ssum = s;
    }
    __syncthreads();
    total = ssum;
  } else total = vfix[5];
  const float u = (float)(total / Mglobal);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    vfix[0] = (double)u;
    if (vpart) { vfix[5] = total; if (!(u > 0.0f)) vfix[1] = 0.0; }
  }
  if (!(u > 0.0f)) return;   // q_b ≥ 0: no sample can lose against u ≤ 0
  unsigned long long cnt = 0;
  for (int pos = blockIdx.x * 256 + threadIdx.x; pos < M; pos += gridDim.x * 256) {
    const int smp = perm[pos];
    const float ov = values[smp], R = returns[smp];
    const float cl = fminf(fmaxf(V[pos] - ov, -c.clip), c.clip);
    const float vc = ov + cl;
    const float q = (vc - R) * (vc - R);
    cnt += (u > q) ? 1ull : 0ull;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&vfix[6]), cnt);
  if (vpart) {
    // one GPU, u > 0 (rare): the block that arrives last turns the integer count into vfix[1] and leaves both counters at zero for the next step
    // (integer adds: the count does not depend on the order of arrival)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      unsigned long long* done = reinterpret_cast<unsigned long long*>(&vfix[7]);
      if (atomicAdd(done, 1ull) == (unsigned long long)gridDim.x - 1ull) {
        unsigned long long* tot = reinterpret_cast<unsigned long long*>(&vfix[6]);
        vfix[1] = (double)atomicAdd(tot, 0ull);
        atomicExch(tot, 0ull); atomicExch(done, 0ull);
      }
    }
  }
}
__global__ void wide_vcount_to_double_kernel(double* vfix) {
  if (threadIdx.x == 0 && blockIdx.x == 0) vfix[1] = (double)*reinterpret_cast<unsigned long long*>(&vfix[6]);
}

// ------------------------------------------------------------------------------------------------------
// Loss terms and output cotangents of one minibatch (ppo.jl:213-244; arithmetic identical to update.hip's tile loop)
// ------------------------------------------------------------------------------------------------------
// The loss kernel's view of a sample — five fields of five arrays, fetched through the epoch's permutation — as ONE 32-byte piece, packed
// when a buffer field changed (h->recs_dirty, the flag the fused path's 64-byte records use): five scattered 4-byte reads dragged five
// sectors per sample, 64 µs of the 0.52 M-sample launch.
__global__ void __launch_bounds__(256) wide_pack_rec_kernel(int B, const int32_t* __restrict__ action, const float* __restrict__ logprob,
                                                           const float* __restrict__ value, const float* __restrict__ adv,
                                                           const float* __restrict__ ret, float* __restrict__ rec) {
  for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
    f32x4 r0, r1 = {0.0f, 0.0f, 0.0f, 0.0f};
    r0[0] = __int_as_float(action[b]); r0[1] = logprob[b]; r0[2] = value[b]; r0[3] = adv[b]; r1[0] = ret[b];
    reinterpret_cast<f32x4*>(rec)[2 * (size_t)b] = r0;
    reinterpret_cast<f32x4*>(rec)[2 * (size_t)b + 1] = r1;
  }
}
struct WLossArgs {
  DevCfg c; const int32_t* perm; float* Z; int A8; const float* V; float* dv8;
  const float* rec;
  const double* adv_ms; int mb; const double* vfix; double Mglobal; double* lpart;
};

__global__ void __launch_bounds__(256) wide_loss_kernel(WLossArgs a) {
  const DevCfg& c = a.c;
  const int A = c.A;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  const bool ok = pos < c.M;
  double sums[4] = {0.0, 0.0, 0.0, 0.0};
  float dout[AMAX];
#pragma unroll
  for (int i = 0; i < AMAX; ++i) dout[i] = 0.0f;
  float dvf = 0.0f;
  if (ok) {
    const int smp = a.perm[pos];
    const double invM = 1.0 / a.Mglobal;
    // policy loss + entropy (ppo.jl:213,219-228,242)
    float z[AMAX], pr[AMAX], lp[AMAX];
    load_logits(a.Z, a.A8, A, (size_t)pos, z);
    softmax_rt(z, A, pr, lp);
    const f32x4 r0 = reinterpret_cast<const f32x4*>(a.rec)[2 * (size_t)smp], r1 = reinterpret_cast<const f32x4*>(a.rec)[2 * (size_t)smp + 1];
    const int act = __float_as_int(r0[0]);
    const float nlp = pick_rt(lp, A, act);
    double Hs = 0.0;
#pragma unroll
    for (int i = 0; i < AMAX; ++i) if (i < A) Hs += (double)(-(pr[i] * lp[i]));
    const float mean_f = (float)a.adv_ms[2 * a.mb];
    const double inv_denom = 1.0 / ((double)(float)a.adv_ms[2 * a.mb + 1] + 1e-8);
    const float eps = c.clip, lo = 1.0f - c.clip, hi = 1.0f + c.clip;
    const double Ahat = (double)(r0[3] - mean_f) * inv_denom;
    const float ratio = expf(nlp - r0[1]);
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const double pg1 = -Ahat * (double)ratio, pg2 = -Ahat * (double)rc;
    double dnlp, pg;
    if (pg1 > pg2) { pg = pg1; dnlp = pg1; }
    else { pg = pg2; dnlp = (ratio >= lo && ratio <= hi) ? pg1 : 0.0; }
    dnlp *= invM;
    const double entk = (double)c.ent_coeff / ((double)A * a.Mglobal);
#pragma unroll
    for (int i = 0; i < AMAX; ++i)
      if (i < A) dout[i] = (float)(dnlp * ((i == act ? 1.0 : 0.0) - (double)pr[i]) + entk * (double)pr[i] * ((double)lp[i] + Hs));
    // value loss (ppo.jl:214,231-240)
    const float v = a.V[pos], R = r1[0], ov = r0[2];
    const double vk = (double)c.v_coef * 0.5 * invM;
    double dv, term;
    if (c.clip_vloss) {
      const float u = (float)a.vfix[0];
      const double nwin = a.vfix[1];
      const float dvv = v - ov;
      const float cl = fminf(fmaxf(dvv, -eps), eps);
      const float vc = ov + cl;
      const float q = (vc - R) * (vc - R);
      const bool q_wins = !(u > q);  // max.(u, q): ties → q
      term = q_wins ? (double)q : (double)u;
      const double inner = (q_wins && dvv >= -eps && dvv <= eps) ? 2.0 * (double)(vc - R) : 0.0;
      dv = vk * (nwin * invM + inner);
    } else {
      const float e = v - R;
      term = (double)(e * e);
      dv = vk * 2.0 * (double)e;
    }
    dvf = (float)dv;
    sums[0] = pg; sums[1] = Hs; sums[2] = (double)(v - R * R); sums[3] = term;
    // cotangents, K padded to a multiple of 8 with zeros for the head's backward GEMM
#pragma unroll
    for (int i = 0; i < AMAX; ++i) if (i < a.A8) a.Z[(size_t)a.A8 * pos + i] = dout[i];   // dout[i] = 0 for i ≥ A
    f32x4 d0 = {dvf, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
    reinterpret_cast<f32x4*>(a.dv8)[2 * (size_t)pos] = d0;
    reinterpret_cast<f32x4*>(a.dv8)[2 * (size_t)pos + 1] = d1;
  }
  // block partials: 4 loss sums, db3 of the actor (Σ dZ) and of the critic (Σ dv)
  __shared__ double sm[4][WLS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const double s = wave_sum(sums[q]); if (lane == 0) sm[wave][q] = s; }
#pragma unroll
  for (int i = 0; i < AMAX; ++i) {
    if (i < A) { const double s = wave_sum((double)dout[i]); if (lane == 0) sm[wave][4 + i] = s; }
  }
  { const double s = wave_sum((double)dvf); if (lane == 0) sm[wave][4 + AMAX] = s; }
  __syncthreads();
  if (threadIdx.x < 4 + AMAX + 1) {
    const int q = threadIdx.x;
    const bool live = q < 4 || q == 4 + AMAX || (q - 4) < A;
    a.lpart[(size_t)blockIdx.x * WLS + q] = live ? (sm[0][q] + sm[1][q]) + (sm[2][q] + sm[3][q]) : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------
// Fixed-order sum of all partials → flat Flux-ordered gradient + the four loss sums (the all-reduce message)
// ------------------------------------------------------------------------------------------------------
struct WRedArgs {
  const float* part[12]; int nparts[12]; int off[13];
  int boff[13];                      // first block of every array (256 elements per block; arrays without partials get none)
  const double* lpart; int nlb; float* out; int P; int A;
  int with_stats; StatsArgs st;      // one GPU: the block that folds the loss sums also writes the "Training Statistics" record (no launch of its own)
};

__global__ void __launch_bounds__(256) wide_reduce_kernel(WRedArgs a) {
  if (blockIdx.x == gridDim.x - 1) {
    // loss sums and head-bias gradients: 4 + 16 + 1 quantities over nlb block partials — all of them through ONE tree (the same per-thread sums and the same
    // tree per quantity as when they were done one after the other: nine trees of eight barriers each were this kernel's critical path, ~25 of its 40 µs)
    constexpr int NQ = 4 + AMAX + 1;
    __shared__ double sm[NQ][256];
    __shared__ float lsum4[4];
    double s[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q] = 0.0;
    for (int i = threadIdx.x; i < a.nlb; i += 256) {
      const double* row = a.lpart + (size_t)i * WLS;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool live = q < 4 || q == 4 + AMAX || (q - 4) < a.A;
        if (live) s[q] += row[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) sm[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
      if ((int)threadIdx.x < w) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const bool live = q < 4 || q == 4 + AMAX || (q - 4) < a.A;
          if (live) sm[q][threadIdx.x] += sm[q][threadIdx.x + w];
        }
      }
      __syncthreads();
    }
    if (threadIdx.x < NQ) {
      const int q = threadIdx.x;
      const bool live = q < 4 || q == 4 + AMAX || (q - 4) < a.A;
      if (live) {
        const float f = (float)sm[q][0];
        if (q < 4) { a.out[a.P + q] = f; lsum4[q] = f; }
        else if (q == 4 + AMAX) a.out[a.off[11]] = f;
        else a.out[a.off[5] + (q - 4)] = f;
      }
    }
    __syncthreads();
    if (a.with_stats && threadIdx.x == 0)
      compute_stats4(lsum4[0], lsum4[1], lsum4[2], lsum4[3], a.st.c, a.st.Mglobal, a.st.adv_ms, a.st.mb, a.st.vfix, a.st.out, 1);
    return;
  }
  // 256 elements of ONE array per block, 4 threads per element quad: thread (e4, pg) sums partials pg, pg + 4, … of four consecutive elements (16-byte
  // loads: the kernel moves 67 MB at C3 and ran at 1.7 TB/s with 4-byte loads); folded in pg order — per element the same sequence of additions as before
  __shared__ double fold[4][256];
  const int e4 = threadIdx.x & 63, pg = threadIdx.x >> 6;
  int arr = 0;
#pragma unroll
  for (int k = 1; k < 12; ++k) arr = ((int)blockIdx.x >= a.boff[k]) ? k : arr;
  const int size = a.off[arr + 1] - a.off[arr], idx = ((int)blockIdx.x - a.boff[arr]) * 256 + 4 * e4, np = a.nparts[arr];
  const float* p = a.part[arr];
  const bool live = p != nullptr && idx < size;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (live) {
    if ((size & 3) == 0) {
      int q = pg;
      for (; q + 4 < np; q += 8) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)q * size + idx), v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(q + 4) * size + idx);
        s0 += (double)v0[0]; s1 += (double)v0[1]; s2 += (double)v0[2]; s3 += (double)v0[3];
        s0 += (double)v1[0]; s1 += (double)v1[1]; s2 += (double)v1[2]; s3 += (double)v1[3];
      }
      for (; q < np; q += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)q * size + idx);
        s0 += (double)v0[0]; s1 += (double)v0[1]; s2 += (double)v0[2]; s3 += (double)v0[3];
      }
    } else {
      for (int q = pg; q < np; q += 4) {
        const float* r = p + (size_t)q * size + idx;
        s0 += (double)r[0];
        if (idx + 1 < size) s1 += (double)r[1];
        if (idx + 2 < size) s2 += (double)r[2];
        if (idx + 3 < size) s3 += (double)r[3];
      }
    }
  }
  fold[pg][4 * e4] = s0; fold[pg][4 * e4 + 1] = s1; fold[pg][4 * e4 + 2] = s2; fold[pg][4 * e4 + 3] = s3;
  __syncthreads();
  {
    const int e = threadIdx.x, i = ((int)blockIdx.x - a.boff[arr]) * 256 + e;
    if (p != nullptr && i < size) a.out[a.off[arr] + i] = (float)(((fold[0][e] + fold[1][e]) + fold[2][e]) + fold[3][e]);
  }
}

__global__ void wide_stats_kernel(const float* __restrict__ msg, int P, StatsArgs st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  compute_stats(msg, P, st.c, st.Mglobal, st.adv_ms, st.mb, st.vfix, st.out, 1);
}

// ------------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------------
int wide_policy_act(crl_ppo* h, const float* obs_d, const double* u_d, int n, int32_t* action_d, float* logprob_d, float* value_d) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (ensure_pack(h)) return 1;
  for (int o = 0; o < n; o += w->Mw) {
    const int m = (n - o) < w->Mw ? (n - o) : w->Mw;
    if (wide_forward(h, 0, obs_d + (size_t)w->D * o, w->D, nullptr, m, w->z, w->A8)) return 1;
    if (value_d && wide_forward(h, 1, obs_d + (size_t)w->D * o, w->D, nullptr, m, w->v, 1, true)) return 1;
    hipLaunchKernelGGL(wide_sample_kernel, dim3((m + 255) / 256), dim3(256), 0, h->stream, w->z, w->A8, w->A, w->v, u_d + o, m,
                       action_d + o, logprob_d + o, value_d ? value_d + o : nullptr);
    CRL_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

int wide_logprob_actions(crl_ppo* h, const float* obs_d, const int32_t* act_d, int n, float* logprob_d, float* ent_d) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (ensure_pack(h)) return 1;
  for (int o = 0; o < n; o += w->Mw) {
    const int m = (n - o) < w->Mw ? (n - o) : w->Mw;
    if (wide_forward(h, 0, obs_d + (size_t)w->D * o, w->D, nullptr, m, w->z, w->A8)) return 1;
    hipLaunchKernelGGL(wide_logprob_kernel, dim3((m + 255) / 256), dim3(256), 0, h->stream, w->z, w->A8, w->A, act_d + o, m,
                       logprob_d + o, ent_d + (size_t)w->A * o);
    CRL_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

int wide_next_value(crl_ppo* h) {
  if (ensure_pack(h)) return 1;
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  return wide_forward(h, 1, h->cur_obs, w->D, nullptr, h->dc.nt, h->next_value, 1, true);
}

int wide_env_reset(crl_ppo* h) {
  hipLaunchKernelGGL(wide_synth_reset_kernel, dim3((h->dc.nt + 255) / 256), dim3(256), 0, h->stream, h->dc, h->env_state, h->env_t,
                     h->cur_obs, h->next_done, h->ep_return, h->ep_length, h->ep_stats);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

static bool wide_fused_ok(const crl_ppo* h);
int wide_rollout(crl_ppo* h) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (ensure_pack(h)) return 1;
  CRL_HIP_CHECK(hipMemsetAsync(h->ep_stats, 0, 4 * sizeof(double), h->stream));
  WStepArgs a;
  a.c = h->dc; a.Z = w->z; a.A8 = w->A8; a.V = w->v;
  a.obs = h->obs; a.action = h->action; a.logprob = h->logprob; a.reward = h->reward; a.terminal = h->terminal; a.value = h->value;
  a.env_state = h->env_state; a.env_t = h->env_t; a.cur_obs = h->cur_obs; a.next_done = h->next_done;
  a.ep_return = h->ep_return; a.ep_length = h->ep_length; a.ep_stats = h->ep_stats; a.iteration = (uint64_t)h->iteration;
  a.ring = h->ep_ring; a.ring_count = h->ep_ring_count; a.ring_cap = h->ep_ring_cap;
  if (h->ep_ring_cap > 0) CRL_HIP_CHECK(hipMemsetAsync(h->ep_ring_count, 0, sizeof(uint32_t), h->stream));
  ProfScope ps(h, CRL_K_ROLLOUT);
  if ((opt(h, OPT_WIDE_RS) & 2) && w->H == 256 && wide_x2(h) && w->D <= 16 && w->A <= 8 && opt(h, OPT_WIDE_ROLLOUT_PERSIST) >= 2 && h->dc.nt % RR_MB == 0 &&
      w->lds_max >= RR_LDS && wide_fused_ok(h) && h->dc.B % FX_MB == 0) {
    // register-stationary actor for all steps (wide_rs_rollout_kernel), then the critic over the stored observations as ONE batched forward
    RsRollArgs r;
    {
      const NetOff o = net_off(256, w->D, w->A);
      const float* pk = w->pack + w->pk_base[0];
      RollPCNet& n = r.n;
      n.W1f = pk + w->pk[0].w1f; n.w1sc = w->wsc + 4; n.Wx2 = pk + w->pk[0].x2f; n.b2 = h->params + o.b2; n.wsc = w->wsc;
      n.W3t = pk + w->pk[0].w3t; n.b3 = h->params + o.b3; n.Z = w->z; n.A = w->A; n.ldz = w->A8; n.rat = 1;
    }
    r.s = a; r.s.V = nullptr; r.D = w->D;
    const int nb = h->dc.nt / RR_MB;
#define CRL_RSROLL(dp, na) hipLaunchKernelGGL((wide_rs_rollout_kernel<dp, na>), dim3(nb), dim3(512), RR_LDS, h->stream, r)
    if (w->D8 == 8 && w->A <= 4) CRL_RSROLL(8, 4);
    else if (w->D8 == 8) CRL_RSROLL(8, 8);
    else if (w->A <= 4) CRL_RSROLL(16, 4);
    else CRL_RSROLL(16, 8);
#undef CRL_RSROLL
    CRL_HIP_CHECK(hipGetLastError());
    // values of all num_steps x num_envs stored observations (ppo.jl:128 evaluates the critic on the same observation the buffer keeps)
    FusedFwdPCArgs q;
    {
      const NetOff o = net_off(256, w->D, 1);
      const float* P = h->params + h->Pa;
      const float* pk = w->pack + w->pk_base[1];
      q.obs = h->obs; q.perm = nullptr; q.D = w->D; q.W1f = pk + w->pk[1].w1f; q.w1sc = w->wsc + 4 + 2; q.Wx2 = pk + w->pk[1].x2f; q.b2 = P + o.b2;
      q.wsc = w->wsc + 2; q.W3t = pk + w->pk[1].w3t; q.b3 = P + o.b3; q.A = 1; q.ldz = 1; q.H1 = nullptr; q.H2 = nullptr; q.Z = h->value; q.M = h->dc.B;
    }
    if (!(opt(h, OPT_WIDE_RS) & 4) || w->D % 4 != 0 || w->lds_max < R2_LDS) {  // (bit 2: the same pass on the register-stationary forward instead — measured slower: 1.21 vs 0.87 ms at C3)
      int nbc = w->cus; const int ntiles = h->dc.B / FX_MB; if (nbc > ntiles) nbc = ntiles;
      if (w->D8 == 8) hipLaunchKernelGGL((wide_fused_fwd_pc_kernel<8, false, 2>), dim3(nbc, 1), dim3(512), pc_lds(2), h->stream, q, q);
      else hipLaunchKernelGGL((wide_fused_fwd_pc_kernel<16, false, 2>), dim3(nbc, 1), dim3(512), pc_lds(2), h->stream, q, q);
    } else {
      // register-stationary forward, every block on the critic: the grid's y = 0 half (the actor flavour) gets an empty argument and leaves at once
      FusedFwdPCArgs q0 = q; q0.M = 0;
      int nbc = w->cus; const int nt32 = h->dc.B / RS_MB; if (nbc > nt32) nbc = nt32;
      if (w->D8 == 8) hipLaunchKernelGGL((wide_rs_fwd_kernel<8, 4, false>), dim3(nbc, 2), dim3(512), R2_LDS, h->stream, q0, q);
      else hipLaunchKernelGGL((wide_rs_fwd_kernel<16, 4, false>), dim3(nbc, 2), dim3(512), R2_LDS, h->stream, q0, q);
    }
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (w->H == 256 && wide_x2(h) && w->D <= 16 && w->A <= PC_AMAX && opt(h, OPT_WIDE_ROLLOUT_PERSIST) >= 2 && h->dc.nt % RP_MB == 0 && w->lds_max >= RP_LDS) {
    // one launch for all steps, producer / consumer form (wide_rollout_pc_kernel)
    RollPCArgs r;
    const bool fast_ok = !opt(h, OPT_WIDE_TANH_RATIONAL);
    for (int net = 0; net < 2; ++net) {
      const int NO = net ? 1 : w->A;
      const NetOff o = net_off(256, w->D, NO);
      const float* P = h->params + (net ? h->Pa : 0);
      const float* pk = w->pack + w->pk_base[net];
      RollPCNet& n = r.n[net];
      n.W1f = pk + w->pk[net].w1f; n.w1sc = w->wsc + 4 + 2 * net; n.Wx2 = pk + w->pk[net].x2f; n.b2 = P + o.b2; n.wsc = w->wsc + 2 * net;
      n.W3t = pk + w->pk[net].w3t; n.b3 = P + o.b3; n.Z = net ? w->v : w->z; n.A = NO; n.ldz = net ? 1 : w->A8;
      n.rat = (net == 1 && fast_ok) ? 0 : 1;            // the actor keeps tanh_fast: its logits decide bit-compared action indices
    }
    r.s = a; r.D = w->D;
    const int nb = h->dc.nt / RP_MB;
    if (w->D8 == 8) hipLaunchKernelGGL(wide_rollout_pc_kernel<8>, dim3(nb), dim3(512), RP_LDS, h->stream, r);
    else hipLaunchKernelGGL(wide_rollout_pc_kernel<16>, dim3(nb), dim3(512), RP_LDS, h->stream, r);
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (w->H == 256 && wide_x2(h) && w->D <= 16 && opt(h, OPT_WIDE_ROLLOUT_PERSIST)) {   // one launch for all steps (wide_rollout_persist_kernel)
    WRollArgs r;
    const bool fast_ok = !opt(h, OPT_WIDE_TANH_RATIONAL);
    for (int net = 0; net < 2; ++net) {
      const int NO = net ? 1 : w->A;
      const NetOff o = net_off(256, w->D, NO);
      const float* P = h->params + (net ? h->Pa : 0);
      const float* pk = w->pack + w->pk_base[net];
      DenseX3Args& x = r.x[net];
      x.Wx3 = pk + w->pk[net].x2f; x.X = w->h1[net]; x.K = 256; x.bias = P + o.b2; x.S = nullptr; x.Y = w->h2[net]; x.M = h->dc.nt;
      x.W3t = pk + w->pk[net].w3t; x.b3 = P + o.b3; x.Z = net ? w->v : w->z; x.A = NO; x.ldz = net ? 1 : w->A8;
      x.dZ = nullptr; x.ldd = 0; x.Ad = 0; x.bz = nullptr; x.bld = 0; x.bA = 0; x.wmax = nullptr;
      x.fast_act = (net == 1 && fast_ok) ? 1 : 0;      // the actor keeps tanh_fast: its logits decide bit-compared action indices
      x.wsc = w->wsc + 2 * net;
      r.W1[net] = P + o.W1; r.b1[net] = P + o.b1; r.h1[net] = w->h1[net]; r.fast[net] = x.fast_act;
    }
    r.s = a; r.D = w->D;
    const size_t s0 = dense_x2_smem(r.x[0], 32), s1 = dense_x2_smem(r.x[1], 32);
    r.obs_off = ((s0 > s1 ? s0 : s1) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(wide_rollout_persist_kernel, dim3((h->dc.nt + 31) / 32), dim3(512), r.obs_off + (size_t)32 * w->D * 4, h->stream, r);
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  for (int step = 0; step < h->dc.k; ++step) {
    if (wide_forward_pair(h, h->cur_obs, w->D, h->dc.nt, w->z, w->A8, w->v, 1)) return 1;   // ppo.jl:127-128
    a.step = step;
    hipLaunchKernelGGL(wide_step_kernel, dim3((h->dc.nt + 255) / 256), dim3(256), 0, h->stream, a);
    CRL_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// backward of one network: head cotangent (K padded to 8, ld ldd) → all six parameter-gradient partials
static int wide_backward(crl_ppo* h, int net, const float* dOut, int ldd, const int32_t* idx) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  const int H = w->H, NO = net ? 1 : w->A, M = h->dc.M;
  const NetOff o = net_off(H, w->D, NO);
  (void)o;
  const float* pk = w->pack + w->pk_base[net];
  // δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²) comes out of the dW3 sweep over h2 (wide_skinny_kernel<.., D2>) when the head has ≤ 8 outputs,
  // else it is its own K ≤ 16 MFMA launch. (Measured and dropped: forming δ2 on the fly inside its two consumers — update 76.4 vs
  // 70.2 ms per iteration at C3, the extra VALU work sits on those kernels' critical path.)
  constexpr bool fuse2 = false;
  const bool d2_sweep = NO <= 8;
  DenseArgs d;
  d.idx = nullptr; d.bias = nullptr; d.M = M;
  if (!fuse2 && !d2_sweep) {
    d.W = pk + w->pk[net].w3t; d.Kp = ldd; d.X = dOut; d.ldx = ldd; d.Kt = ldd; d.S = w->h2[net]; d.lds = H; d.Y = w->dA; d.ldy = H; d.Nt = H;
    if (dense_launch<EPI_DTANH>(h->stream, H, d)) return 1;
  }
  // dW3[a, k] = Σ δ3[a]·h2[k]
  SkinnyArgs s;
  s.Big = w->h2[net]; s.H = H; s.Small = dOut; s.lds = ldd; s.idx = nullptr; s.M = M; s.chunk = w->chunks;
  s.pW = w->pW3[net]; s.os_row = NO; s.os_s = 1; s.St = NO; s.wsize = H * NO; s.pB = nullptr;
  s.D2out = d2_sweep ? w->dA : nullptr; s.W3t = pk + w->pk[net].w3t;
  if (skinny_launch(h->stream, w->Ss, s)) return 1;
  s.D2out = nullptr;
  // dW2 = δ2·h1ᵀ, db2 = Σ δ2
  WgradArgs g;
  g.dY = fuse2 ? w->h2[net] : w->dA; g.X = w->h1[net]; g.H = H; g.M = M; g.chunk = w->chunk2; g.pW = w->pW2[net]; g.pB = w->pB2[net];
  g.dZ = fuse2 ? dOut : nullptr; g.ldd = ldd; g.Ad = NO; g.W3t = pk + w->pk[net].w3t;
  g.bz = dOut; g.bld = ldd; g.bA = NO; g.wmax = pk + w->pk[net].wmax;
  const bool x2 = H == 256 && wide_x2(h);
  if (x2) hipLaunchKernelGGL(wide_wgrad_x2_kernel<4>, dim3(w->S2, 2), dim3(512), 2 * (256 + 128) * X3ROW * 2, h->stream, g);
  else if (H == 256 && wide_x3(h))   // 256×128 output tiles (dY read twice, X once: 1.5 GB per launch at C3; 128×128 tiles read 2 GB)
    hipLaunchKernelGGL(wide_wgrad_x3_kernel<4>, dim3(w->S2, 2), dim3(512), 3 * (256 + 128) * X3ROW * 2, h->stream, g);
  else if (H >= 128) { const int nb = H / 128; hipLaunchKernelGGL(wide_wgrad_kernel<2>, dim3(w->S2, nb * nb), dim3(256), 0, h->stream, g); }
  else hipLaunchKernelGGL(wide_wgrad_kernel<1>, dim3(w->S2, 1), dim3(256), 0, h->stream, g);
  CRL_HIP_CHECK(hipGetLastError());
  // δ1 = (W2ᵀ·δ2) ⊙ (1 − h1²)
  if (H == 256 && wide_x3(h)) {
    DenseX3Args x;
    x.Wx3 = pk + w->pk[net].x3b; x.X = fuse2 ? w->h2[net] : w->dA; x.K = H; x.bias = nullptr; x.S = w->h1[net]; x.Y = w->dB; x.M = M;
    x.W3t = pk + w->pk[net].w3t; x.b3 = nullptr; x.Z = nullptr; x.A = 0; x.ldz = 0;
    x.dZ = fuse2 ? dOut : nullptr; x.ldd = ldd; x.Ad = NO;
    x.bz = dOut; x.bld = ldd; x.bA = NO; x.wmax = pk + w->pk[net].wmax;
    if (x2) { x.Wx3 = pk + w->pk[net].x2b; x.wsc = w->wsc + 2 * net; if (dense_x2_launch<EPI_DTANH>(h->stream, x)) return 1; }
    else if (dense_x3_launch<EPI_DTANH>(h->stream, x)) return 1;
  } else {
    d.W = pk + w->pk[net].w2t; d.Kp = H; d.X = w->dA; d.ldx = H; d.Kt = H; d.S = w->h1[net]; d.lds = H; d.Y = w->dB; d.ldy = H; d.Nt = H;
    if (dense_launch<EPI_DTANH>(h->stream, H, d)) return 1;
  }
  // dW1 = δ1·xᵀ, db1 = Σ δ1
  s.Big = w->dB; s.Small = h->obs; s.lds = w->D; s.idx = idx; s.pW = w->pW1[net]; s.os_row = 1; s.os_s = H; s.St = w->D;
  s.wsize = H * w->D; s.pB = w->pB1[net];
  return skinny_launch(h->stream, w->Ss, s);
}

// The update pass's forward of BOTH networks as one launch of the tile-resident kernel (wide_fused.hpp): 2x256, fp16x2, obs_dim <= 16,
// exp2-based activation (what wide_forward(…, fast_act = true) computes layer by layer). Option wide_fuse = 0 keeps the layer-wise launches.
static bool wide_fused_ok(const crl_ppo* h) {
  const WideWs* w = static_cast<const WideWs*>(h->wide_ws);
  return w->H == 256 && wide_x2(h) && w->D8 <= 16 && w->A <= AMAX && opt(h, OPT_WIDE_FUSE) != 0 && !opt(h, OPT_WIDE_TANH_RATIONAL) &&
         w->lds_max >= FB_OFF_W3 + FB_AMAX * 1024;    // the largest request of the fused forward / backward / weight-gradient kernels
}
// h1 is never stored: the fused backward does not read it and the weight gradient regenerates it (option wide_fuse = 3, the default)
static bool wide_h1_free(const crl_ppo* h) {
  const WideWs* w = static_cast<const WideWs*>(h->wide_ws);
  return opt(h, OPT_WIDE_FUSE) >= 3 && opt(h, OPT_WIDE_FUSE_PC) && w->A <= PC_AMAX && w->A <= FB_AMAX && w->D <= 16;
}
static int wide_forward_fused(crl_ppo* h, const int32_t* perm, int M) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  FusedFwdArgs a[2];
  for (int net = 0; net < 2; ++net) {
    const int NO = net ? 1 : w->A;
    const NetOff o = net_off(256, w->D, NO);
    const float* P = h->params + (net ? h->Pa : 0);
    const float* pk = w->pack + w->pk_base[net];
    a[net].obs = h->obs; a[net].perm = perm; a[net].D = w->D;
    a[net].W1s = pk + w->pk[net].w1s; a[net].Wx2 = pk + w->pk[net].x2f; a[net].b2 = P + o.b2; a[net].wsc = w->wsc + 2 * net;
    a[net].W3t = pk + w->pk[net].w3t; a[net].b3 = P + o.b3; a[net].A = NO; a[net].ldz = net ? 1 : w->A8;
    a[net].H1 = w->h1[net]; a[net].H2 = w->h2[net]; a[net].Z = net ? w->v : w->z; a[net].M = M;
  }
  if (opt(h, OPT_WIDE_FUSE_PC) && w->A <= PC_AMAX) {
    // producer / consumer form, persistent: one block per CU, half of them per network
    FusedFwdPCArgs q[2];
    for (int net = 0; net < 2; ++net) {
      const float* pk = w->pack + w->pk_base[net];
      q[net].obs = a[net].obs; q[net].perm = perm; q[net].D = w->D; q[net].W1f = pk + w->pk[net].w1f; q[net].w1sc = w->wsc + 4 + 2 * net;
      q[net].Wx2 = a[net].Wx2; q[net].b2 = a[net].b2; q[net].wsc = a[net].wsc; q[net].W3t = a[net].W3t; q[net].b3 = a[net].b3; q[net].A = a[net].A;
      q[net].ldz = a[net].ldz; q[net].H1 = a[net].H1; q[net].H2 = a[net].H2; q[net].Z = a[net].Z; q[net].M = M;
    }
    if ((opt(h, OPT_WIDE_RS) & 1) && wide_h1_free(h) && M % RS_MB == 0 && w->D % 4 == 0 && w->lds_max >= R2_LDS) {
      // register-stationary form (wide_rs.hpp): no weight stream at all
      int nbr = w->cus / 2; const int nt32 = M / RS_MB; if (nbr > nt32) nbr = nt32; if (nbr < 1) nbr = 1;
      if (w->D8 == 8 && w->A <= 4) hipLaunchKernelGGL((wide_rs_fwd_kernel<8, 4, true>), dim3(nbr, 2), dim3(512), R2_LDS, h->stream, q[0], q[1]);
      else if (w->D8 == 8) hipLaunchKernelGGL((wide_rs_fwd_kernel<8, 8, true>), dim3(nbr, 2), dim3(512), R2_LDS, h->stream, q[0], q[1]);
      else if (w->A <= 4) hipLaunchKernelGGL((wide_rs_fwd_kernel<16, 4, true>), dim3(nbr, 2), dim3(512), R2_LDS, h->stream, q[0], q[1]);
      else hipLaunchKernelGGL((wide_rs_fwd_kernel<16, 8, true>), dim3(nbr, 2), dim3(512), R2_LDS, h->stream, q[0], q[1]);
      CRL_HIP_CHECK(hipGetLastError());
      return 0;
    }
    int nb = w->cus / 2; const int ntiles = M / FX_MB; if (nb > ntiles) nb = ntiles; if (nb < 1) nb = 1;
    // three weight buffers (the slab after next in flight) where the W3ᵀ table leaves room for them, option wide_fwd_wbufs = 2 keeps two
    const bool three = w->A <= pc_amax(3) && opt(h, OPT_WIDE_FWD_WBUFS) >= 3 && w->lds_max >= pc_lds(3);
    const bool regs = opt(h, OPT_WIDE_FWD_WBUFS) == 0;      // weight fragments straight into the consumers' registers
#define CRL_FWD_PC(dp, wh1)                                                                                                                      \
    do {                                                                                                                                           \
      if (regs) hipLaunchKernelGGL((wide_fused_fwd_pc_kernel<dp, wh1, 0>), dim3(nb, 2), dim3(512), pc_lds(0), h->stream, q[0], q[1]);              \
      else if (three) hipLaunchKernelGGL((wide_fused_fwd_pc_kernel<dp, wh1, 3>), dim3(nb, 2), dim3(512), pc_lds(3), h->stream, q[0], q[1]);        \
      else hipLaunchKernelGGL((wide_fused_fwd_pc_kernel<dp, wh1, 2>), dim3(nb, 2), dim3(512), pc_lds(2), h->stream, q[0], q[1]);                   \
    } while (0)
    if (wide_h1_free(h)) { if (w->D8 == 8) CRL_FWD_PC(8, false); else CRL_FWD_PC(16, false); }
    else if (w->D8 == 8) CRL_FWD_PC(8, true);
    else CRL_FWD_PC(16, true);
#undef CRL_FWD_PC
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  const dim3 grid((M + FX_MB - 1) / FX_MB, 2);
  if (w->D8 == 8) hipLaunchKernelGGL((wide_fused_fwd_kernel<8, true>), grid, dim3(512), FX_LDS, h->stream, a[0], a[1]);
  else hipLaunchKernelGGL((wide_fused_fwd_kernel<16, true>), grid, dim3(512), FX_LDS, h->stream, a[0], a[1]);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// Backward of BOTH networks: one launch of the tile-resident kernel (δ2 → dA / dB, dW1 / db1 partials), then per network the dW3 sweep over
// h2 and the 256x256 weight-gradient kernel on the stored δ2 and h1.
// the register-stationary backward with dW3 inside (wide_rs bits 3 + 4) will run for this minibatch: nothing is launched on the side stream
static bool wide_rs_bwd_forms_dw3(const crl_ppo* h, int M) {
  const WideWs* w = static_cast<const WideWs*>(h->wide_ws);
  const bool split = wide_h1_free(h) && opt(h, OPT_WIDE_WGRAD_FULL) && opt(h, OPT_WIDE_D2_SPLIT) && w->chunk2 % 32 == 0 && w->lds_max >= WS_LDS;
  return split && (opt(h, OPT_WIDE_RS) & 8) && (opt(h, OPT_WIDE_RS) & 16) && M % RS_MB == 0 && w->A <= 8 && w->D % 4 == 0 && w->lds_max >= RB_LDS;
}
static int wide_backward_fused(crl_ppo* h, const int32_t* perm, int M) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  const int ntiles = M / FX_MB;
  // δ2 as the backward's own fp16x2 pieces when its only reader is the 256x256 weight-gradient kernel (whole 32-sample slabs: M % 128 == 0 here)
  const bool split = wide_h1_free(h) && opt(h, OPT_WIDE_WGRAD_FULL) && opt(h, OPT_WIDE_D2_SPLIT) && w->chunk2 % 32 == 0 && w->lds_max >= WS_LDS;
  int nb = w->cus / 2; if (nb > ntiles) nb = ntiles; if (nb > w->Ss) nb = w->Ss; if (nb < 1) nb = 1;
  FusedBwdArgs a[2];
  for (int net = 0; net < 2; ++net) {
    const int NO = net ? 1 : w->A;
    const float* pk = w->pack + w->pk_base[net];
    a[net].H2 = w->h2[net]; a[net].dZ = net ? w->dv8 : w->z; a[net].ldd = net ? 8 : w->A8; a[net].A = NO;
    a[net].W3t = pk + w->pk[net].w3t; a[net].wmax = pk + w->pk[net].wmax; a[net].Wx2b = pk + w->pk[net].x2b; a[net].wsc = w->wsc + 2 * net;
    a[net].obs = h->obs; a[net].perm = perm; a[net].D = w->D; a[net].W1s = pk + w->pk[net].w1s;
    a[net].D2 = net ? w->dB : w->dA; a[net].pW1 = w->pW1[net]; a[net].pB1 = w->pB1[net]; a[net].M = M;
    a[net].D2h = split ? reinterpret_cast<_Float16*>(net ? w->dB : w->dA) : nullptr; a[net].d2s = split ? w->d2s + (size_t)net * w->Mw : nullptr;
  }
  const size_t lds = (size_t)FB_OFF_W3 + (size_t)w->A * 1024;
  const bool rsb = split && (opt(h, OPT_WIDE_RS) & 8) && M % RS_MB == 0 && w->A <= 8 && w->D % 4 == 0 && w->lds_max >= RB_LDS;
  if (rsb) {     // register-stationary form (wide_rs.hpp)
    for (int net = 0; net < 2; ++net) { const float* pk = w->pack + w->pk_base[net]; a[net].W1f = pk + w->pk[net].w1f; a[net].w1sc = w->wsc + 4 + 2 * net; a[net].pW3 = w->pW3[net]; }
    int nbr = w->cus / 2; const int nt32 = M / RS_MB; if (nbr > nt32) nbr = nt32; if (nbr > w->Ss) nbr = w->Ss; if (nbr < 1) nbr = 1;
    nb = nbr;
    w->fb_blocks_net[0] = w->fb_blocks_net[1] = nbr;
    const int pct = (int)opt(h, OPT_WIDE_RS_ACTOR_PCT);
    if (pct != 50 && 2 * nbr == w->cus && w->cus <= 2 * w->Ss) {        // uneven split of the CUs: the grid's x extent is the larger share
      int na = w->cus * pct / 100; if (na < 1) na = 1; if (na > w->cus - 1) na = w->cus - 1;
      if (na <= w->Ss && w->cus - na <= w->Ss) { a[0].nblk = na; a[1].nblk = w->cus - na; w->fb_blocks_net[0] = na; w->fb_blocks_net[1] = w->cus - na; nb = na > w->cus - na ? na : w->cus - na; }
    }
    const bool dw3 = (opt(h, OPT_WIDE_RS) & 16) != 0;                 // dW3 inside the backward kernel instead of the two sweeps over h2
#define CRL_RSB(dp, na)                                                                                                              \
    do {                                                                                                                             \
      if (dw3) hipLaunchKernelGGL((wide_rs_bwd_kernel<dp, na, true>), dim3(nb, 2), dim3(512), RB_LDS, h->stream, a[0], a[1]);        \
      else hipLaunchKernelGGL((wide_rs_bwd_kernel<dp, na, false>), dim3(nb, 2), dim3(512), RB_LDS, h->stream, a[0], a[1]);           \
    } while (0)
    if (w->D8 == 8 && w->A <= 4) CRL_RSB(8, 4);
    else if (w->D8 == 8) CRL_RSB(8, 8);
    else if (w->A <= 4) CRL_RSB(16, 4);
    else CRL_RSB(16, 8);
#undef CRL_RSB
  } else {
#define CRL_BWD(dp, na)                                                                                                                  \
  do {                                                                                                                                   \
    if (split) hipLaunchKernelGGL((wide_fused_bwd_kernel<dp, na, true>), dim3(nb, 2), dim3(512), lds, h->stream, a[0], a[1]);            \
    else hipLaunchKernelGGL((wide_fused_bwd_kernel<dp, na, false>), dim3(nb, 2), dim3(512), lds, h->stream, a[0], a[1]);                 \
  } while (0)
  if (w->D8 == 8 && w->A <= 4) CRL_BWD(8, 4);
  else if (w->D8 == 8) CRL_BWD(8, 8);
  else if (w->A <= 4) CRL_BWD(16, 4);
  else CRL_BWD(16, 8);
#undef CRL_BWD
  }
  CRL_HIP_CHECK(hipGetLastError());
  w->fb_blocks = nb;
  if (!rsb) w->fb_blocks_net[0] = w->fb_blocks_net[1] = nb;
  const bool rsb_dw3 = rsb && (opt(h, OPT_WIDE_RS) & 16) != 0;
  w->w3_blocks = rsb_dw3 ? 1 : 0;
  // The dW3 sweeps over h2 depend on the loss kernel only, like the fused backward: they run on the second stream beside it and join at
  // the end of this function.
  const bool side = opt(h, OPT_SHUFFLE_OVERLAP) != 0 && !rsb_dw3;      // (nothing goes to the side stream when the backward kernel formed dW3 itself: no fork / join — 13 µs per step)
  hipStream_t sk = side ? h->stream2 : h->stream;
  if (side) CRL_HIP_CHECK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));   // ev_fork was recorded behind the loss kernel (wide_grad_passes)
  for (int net = 0; net < 2; ++net) {
    const int NO = net ? 1 : w->A;
    const float* pk = w->pack + w->pk_base[net];
    const float* dOut = net ? w->dv8 : w->z; const int ldd = net ? 8 : w->A8;
    SkinnyArgs s;   // dW3[a, k] = Σ δ3[a]·h2[k]
    s.Big = w->h2[net]; s.H = 256; s.Small = dOut; s.lds = ldd; s.idx = nullptr; s.M = M; s.chunk = w->chunks;
    s.pW = w->pW3[net]; s.os_row = NO; s.os_s = 1; s.St = NO; s.wsize = 256 * NO; s.pB = nullptr; s.D2out = nullptr; s.W3t = pk + w->pk[net].w3t;
    if (!rsb_dw3 && skinny_launch(sk, w->Ss, s)) return 1;
    if (!wide_h1_free(h)) {
      WgradArgs g;    // dW2 = δ2·h1ᵀ, db2 = Σ δ2 from the stored h1
      g.dY = net ? w->dB : w->dA; g.X = w->h1[net]; g.H = 256; g.M = M; g.chunk = w->chunk2; g.pW = w->pW2[net]; g.pB = w->pB2[net];
      g.dZ = nullptr; g.ldd = ldd; g.Ad = NO; g.W3t = pk + w->pk[net].w3t; g.bz = dOut; g.bld = ldd; g.bA = NO; g.wmax = pk + w->pk[net].wmax;
      hipLaunchKernelGGL(wide_wgrad_x2_kernel<4>, dim3(w->S2, 2), dim3(512), 2 * (256 + 128) * X3ROW * 2, h->stream, g);
      CRL_HIP_CHECK(hipGetLastError());
    }
  }
  if (wide_h1_free(h) && !opt(h, OPT_WIDE_WGRAD_FULL)) {
    // h1 regenerated inside wide_wgrad_x2_kernel's own structure (256 x 128 tiles, two blocks per CU: the stream of δ2 hides under the other block)
    for (int net = 0; net < 2; ++net) {
      const float* pk = w->pack + w->pk_base[net];
      WgradArgs g;
      g.dY = net ? w->dB : w->dA; g.X = nullptr; g.H = 256; g.M = M; g.chunk = w->chunk2; g.pW = w->pW2[net]; g.pB = w->pB2[net];
      g.dZ = nullptr; g.ldd = net ? 8 : w->A8; g.Ad = net ? 1 : w->A; g.W3t = pk + w->pk[net].w3t;
      g.bz = net ? w->dv8 : w->z; g.bld = net ? 8 : w->A8; g.bA = net ? 1 : w->A; g.wmax = pk + w->pk[net].wmax;
      g.obs = h->obs; g.perm = perm; g.D = w->D; g.W1f = pk + w->pk[net].w1f; g.w1sc = w->wsc + 4 + 2 * net;
      hipLaunchKernelGGL((wide_wgrad_x2_kernel<4, true>), dim3(w->S2, 2), dim3(512), 2 * (256 + 128) * X3ROW * 2, h->stream, g);
      CRL_HIP_CHECK(hipGetLastError());
    }
  } else if (split) {             // both networks in one launch, 256 x 256 tile per block, from the split planes (wide_wgrad_split_kernel)
    WgradSplitArgs g[2];
    for (int net = 0; net < 2; ++net) {
      const float* pk = w->pack + w->pk_base[net];
      g[net].Yh = reinterpret_cast<const _Float16*>(net ? w->dB : w->dA); g[net].ys = w->d2s + (size_t)net * w->Mw;
      g[net].obs = h->obs; g[net].perm = perm; g[net].D = w->D; g[net].W1f = pk + w->pk[net].w1f; g[net].w1sc = w->wsc + 4 + 2 * net;
      g[net].pW = w->pW2[net]; g[net].pB = w->pB2[net]; g[net].M = M; g[net].chunk = w->chunk2;
    }
    if (w->D8 == 8) hipLaunchKernelGGL((wide_wgrad_split_kernel<8>), dim3(w->S2, 2), dim3(512), WS_LDS, h->stream, g[0], g[1]);
    else hipLaunchKernelGGL((wide_wgrad_split_kernel<16>), dim3(w->S2, 2), dim3(512), WS_LDS, h->stream, g[0], g[1]);
    CRL_HIP_CHECK(hipGetLastError());
  } else if (wide_h1_free(h)) {   // both networks in one launch, 256 x 256 tile per block (wide_wgrad_gen_kernel)
    WgradGenArgs g[2];
    for (int net = 0; net < 2; ++net) {
      const float* pk = w->pack + w->pk_base[net];
      g[net].dY = net ? w->dB : w->dA; g[net].obs = h->obs; g[net].perm = perm; g[net].D = w->D; g[net].W1f = pk + w->pk[net].w1f; g[net].w1sc = w->wsc + 4 + 2 * net;
      g[net].bz = net ? w->dv8 : w->z; g[net].bld = net ? 8 : w->A8; g[net].bA = net ? 1 : w->A; g[net].wmax = pk + w->pk[net].wmax;
      g[net].pW = w->pW2[net]; g[net].pB = w->pB2[net]; g[net].M = M; g[net].chunk = w->chunk2;
    }
    if (w->D8 == 8) hipLaunchKernelGGL((wide_wgrad_gen_kernel<8>), dim3(w->S2, 2), dim3(512), WG_LDS, h->stream, g[0], g[1]);
    else hipLaunchKernelGGL((wide_wgrad_gen_kernel<16>), dim3(w->S2, 2), dim3(512), WG_LDS, h->stream, g[0], g[1]);
    CRL_HIP_CHECK(hipGetLastError());
  }
  // The dW3 partials are first read by the reduction that follows this function: the join sits behind the weight-gradient launches, so the
  // sweeps' blocks may fill the tails of BOTH big kernels (each of those owns every CU's registers while its blocks run; joined before the
  // weight gradient, the two sweeps took 2 x 480 µs on their stream against the backward's 764: the main stream waited for them)
  if (side) { CRL_HIP_CHECK(hipEventRecord(h->ev_join, h->stream2)); CRL_HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_join, 0)); }
  return 0;
}

// forward → u → loss → backward of one minibatch; leaves the per-chunk gradient partials in the workspace
static int wide_grad_passes(crl_ppo* h, int mb, const int32_t* perm, double Mglobal, bool dp) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  const int M = h->dc.M;
  if (wide_fused_ok(h) && M % FX_MB == 0) {
    if (wide_forward_fused(h, perm, M)) return 1;
  } else {
    if (wide_forward(h, 1, h->obs, w->D, perm, M, w->v, 1, true)) return 1;
    if (wide_forward(h, 0, h->obs, w->D, perm, M, w->z, w->A8, true)) return 1;
  }
  if (h->cfg.clip_value_loss) {
    int nb = (M + 255) / 256; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(wide_vsum_kernel, dim3(nb), dim3(256), 0, h->stream, w->v, h->ret, perm, M, w->vpart);
    if (!dp) {
      // one GPU: two launches instead of four — the count kernel folds the partial sums itself and finishes the count in its last block
      hipLaunchKernelGGL(wide_vcount_kernel, dim3(nb), dim3(256), 0, h->stream, h->dc, w->v, h->value, h->ret, perm, M, Mglobal, h->vfix,
                         (const double*)w->vpart, nb);
      CRL_HIP_CHECK(hipGetLastError());
    } else {
      hipLaunchKernelGGL(wide_vsum_final_kernel, dim3(1), dim3(64), 0, h->stream, w->vpart, nb, h->vfix);
      CRL_HIP_CHECK(hipGetLastError());
      if (comm_allreduce(h, h->vfix + 5, 1, true)) return 1;
      hipLaunchKernelGGL(wide_vcount_kernel, dim3(nb), dim3(256), 0, h->stream, h->dc, w->v, h->value, h->ret, perm, M, Mglobal, h->vfix,
                         (const double*)nullptr, 0);
      hipLaunchKernelGGL(wide_vcount_to_double_kernel, dim3(1), dim3(1), 0, h->stream, h->vfix);
      CRL_HIP_CHECK(hipGetLastError());
      if (comm_allreduce(h, h->vfix + 1, 1, true)) return 1;
    }
  }
  {
    WLossArgs a;
    a.c = h->dc; a.perm = perm; a.Z = w->z; a.A8 = w->A8; a.V = w->v; a.dv8 = w->dv8;
    if (h->recs_dirty) {   // a buffer field changed since the records were packed (rollout, GAE, a host write)
      int nbr = (h->dc.B + 255) / 256; if (nbr > 4096) nbr = 4096;
      hipLaunchKernelGGL(wide_pack_rec_kernel, dim3(nbr), dim3(256), 0, h->stream, h->dc.B, h->action, h->logprob, h->value, h->adv, h->ret, w->wrec);
      h->recs_dirty = false;
    }
    a.rec = w->wrec;
    a.adv_ms = h->adv_ms; a.mb = mb; a.vfix = h->vfix; a.Mglobal = Mglobal; a.lpart = w->lpart;
    hipLaunchKernelGGL(wide_loss_kernel, dim3(w->nlb), dim3(256), 0, h->stream, a);
    CRL_HIP_CHECK(hipGetLastError());
  }
  w->fb_blocks = 0; w->w3_blocks = 0;
  if (wide_fused_ok(h) && M % FX_MB == 0 && w->A <= FB_AMAX && opt(h, OPT_WIDE_FUSE) >= 2) {
    if (opt(h, OPT_SHUFFLE_OVERLAP) && !wide_rs_bwd_forms_dw3(h, M)) CRL_HIP_CHECK(hipEventRecord(h->ev_fork, h->stream));   // behind the loss kernel: the side stream starts here
    return wide_backward_fused(h, perm, M);
  }
  if (wide_backward(h, 0, w->z, w->A8, perm)) return 1;
  if (wide_backward(h, 1, w->dv8, 8, perm)) return 1;
  return 0;
}

// One optimiser step's gradient (ppo.jl:197-244): forward → u → loss → backward → fixed-order reduce → [all-reduce] →
// statistics. The gradient message ends up in comm_buf like in update.hip.
int wide_update(crl_ppo* h, int mb, crl_ppo_stats* stats_slot) {
  WideWs* w = static_cast<WideWs*>(h->wide_ws);
  if (ensure_pack(h)) return 1;
  const int M = h->dc.M, P = (int)h->P;
  const int32_t* perm = h->perm + (size_t)mb * M;
  const double Mglobal = (double)M * h->world;
  const bool dp = has_comm(h) || h->external_comm;   // a forced 1-rank communicator still goes through RCCL
  if (h->external_comm && h->world > 1 && h->cfg.clip_value_loss) {
    set_error("wide path: clip_value_loss under host-side exchange (crl_comm_init_external) is not supported; use crl_comm_init");
    return 1;
  }
  {
    ProfScope ps(h, CRL_K_UPDATE);   // HIP events around the whole forward / loss / backward group of this minibatch
    if (wide_grad_passes(h, mb, perm, Mglobal, dp)) return 1;
  }
  {
    WRedArgs r;
    const int H = w->H, D = w->D, A = w->A;
    const int sizes[12] = {H * D, H, H * H, H, A * H, A, H * D, H, H * H, H, H, 1};
    r.off[0] = 0;
    for (int i = 0; i < 12; ++i) r.off[i + 1] = r.off[i] + sizes[i];
    for (int n = 0; n < 2; ++n) {
      const int b = 6 * n;
      r.part[b + 0] = w->pW1[n]; r.nparts[b + 0] = w->fb_blocks ? w->fb_blocks_net[n] : w->Ss;
      r.part[b + 1] = w->pB1[n]; r.nparts[b + 1] = w->fb_blocks ? w->fb_blocks_net[n] : w->Ss;
      r.part[b + 2] = w->pW2[n]; r.nparts[b + 2] = w->S2;
      r.part[b + 3] = w->pB2[n]; r.nparts[b + 3] = w->S2;
      r.part[b + 4] = w->pW3[n]; r.nparts[b + 4] = w->w3_blocks ? w->fb_blocks_net[n] : w->Ss;
      r.part[b + 5] = nullptr; r.nparts[b + 5] = 0;
    }
    r.lpart = w->lpart; r.nlb = w->nlb; r.out = h->comm_buf; r.P = P; r.A = A;
    r.with_stats = dp ? 0 : 1;
    r.st.c = h->dc; r.st.Mglobal = Mglobal; r.st.adv_ms = h->adv_ms; r.st.mb = mb; r.st.vfix = h->vfix; r.st.out = stats_slot; r.st.fused = 0; r.st.dscale = nullptr;
    r.boff[0] = 0;
    for (int i = 0; i < 12; ++i) r.boff[i + 1] = r.boff[i] + (r.part[i] ? (sizes[i] + 255) / 256 : 0);
    ProfScope pr(h, CRL_K_REDUCE);
    hipLaunchKernelGGL(wide_reduce_kernel, dim3(r.boff[12] + 1), dim3(256), 0, h->stream, r);
    CRL_HIP_CHECK(hipGetLastError());
  }
  if (dp) {      // the sums are global only after the all-reduce: the record keeps its own tiny launch
    {
      ProfScope pa(h, CRL_K_ALLREDUCE);
      if (comm_allreduce(h, h->comm_buf, (size_t)P + 4, false)) return 1;
    }
    StatsArgs st;
    st.c = h->dc; st.Mglobal = Mglobal; st.adv_ms = h->adv_ms; st.mb = mb; st.vfix = h->vfix; st.out = stats_slot; st.fused = 0;
    st.dscale = nullptr;
    hipLaunchKernelGGL(wide_stats_kernel, dim3(1), dim3(64), 0, h->stream, h->comm_buf, P, st);
    CRL_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

}  // namespace crl
