// wide_fused.hpp — tile-resident passes of the 2x256 layer-wise path (included by wide.hip; BASELINE config C3: obs 8 / act 4 / 2x256).
//
// The layer-wise kernels of wide.hip stream every [256 x M] activation through HBM between launches (6.4 GB per network and optimiser
// step at M = 524,288) and their slab loops are bound on chip: both operands go global → registers → LDS with two barriers per 32-k slab
// (profiles/r03_c3_pmc_before_tm4.txt: waves parked 64 % of their cycles, matrix pipe busy 0.22). The kernels here keep a 128-sample
// tile's activations on the CU across layers and feed the 256x256 product from two double-buffered LDS streams with ONE barrier per slab:
//   * the weight slab (32 k x 256 rows as fp16x2 A-fragments, 32 KB, already in fragment order in the pack buffer) arrives by LDS-DMA
//     (global_load_lds_dwordx4: no registers, no ds_write, issued a slab ahead);
//   * the activation slab (32 k x 128 samples) is PRODUCED on the CU by the block's 512 threads — thread (sample, 8 consecutive k) —
//     from something small: the forward pass recomputes h1 = tanh(W1·x + b1) from the 8 observation floats of the sample (W1 rows come
//     through scalar loads: they are wave-uniform), so h1 is never read from HBM;
//   * a wave owns a 64-row x 64-sample register tile (2 x 2 MFMA tiles, 64 accumulator registers): 8 fragment reads per 12 MFMAs.
// Products are fp16x2 (mlp_x2.hpp: hi·hi + hi·lo + lo·hi, f32 accumulate), scales as in wide_dense_x2_kernel.
#pragma once

namespace crl {

constexpr int FX_MB = 128;                                   // samples per block tile
constexpr int FX_WBYTES = X2_SLAB_F16 * 2;                   // 32,768: one weight slab
constexpr int FX_XBYTES = 2 * FX_MB * X3ROW * 2;             // 20,480: one activation slab, [piece][sample][X3ROW halves]
constexpr int FX_OFF_X = 2 * FX_WBYTES;
constexpr int FX_LDS = FX_OFF_X + 2 * FX_XBYTES;             // 106,496 bytes: one block per CU
static_assert(8 * 32 * 36 * 4 <= 2 * FX_WBYTES, "epilogue scratch aliases the weight buffers");
static_assert(4 * FX_MB * AMAX * 4 <= 2 * FX_XBYTES, "head partials alias the activation buffers");

typedef const float __attribute__((address_space(4))) cfloat_k;   // constant address space: uniform-address loads become s_load

struct FusedFwdArgs {
  const float* obs; const int32_t* perm; int D;   // sample m's observation: obs + D·(perm ? perm[m] : m)
  const float* W1s;                               // [256][DP] rows of W1·2·log2(e), then b1·2·log2(e) [256] (pack: w1s)
  const float* Wx2;                               // fp16x2 A-fragment slabs of W2·scale (pack: x2f)
  const float* b2; const float* wsc;              // bias of layer 2; {scale, 1/scale} of the W2 pieces
  const float* W3t; const float* b3; int A; int ldz;   // head: W3ᵀ [256 x ·] column-major (ld 256), bias, outputs, ld of Z
  float* H1; float* H2; float* Z; int M;          // H1 may be null (not stored)
};

// one weight slab into LDS by LDS-DMA: 32 pieces of 1 KB, four per wave; the image is the pack buffer's own fragment order
__device__ __forceinline__ void fx_dma_wslab(const float* Wx2, int slab, unsigned char* dst, int wave, int lane) {
  const char* g = reinterpret_cast<const char*>(Wx2) + (size_t)slab * FX_WBYTES;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = i * 8 + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
  }
}

// acc[ai][bi] += W-slab(rows 64rg + 32ai …) · X-slab(samples 64sg + 32bi …) over the slab's two k-steps
__device__ __forceinline__ void fx_compute_slab(const unsigned char* wbuf, const unsigned char* xbuf, int rg, int sg, int lane, f32x16 (&acc)[2][2]) {
  const f16x8* Wl = reinterpret_cast<const f16x8*>(wbuf);
  const _Float16* Xl = reinterpret_cast<const _Float16*>(xbuf);
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    P2 af[2], bf[2];
#pragma unroll
    for (int ai = 0; ai < 2; ++ai) {
      const int fr = (ks * 8 + 2 * rg + ai) * 64 + lane;
      af[ai].hi = Wl[fr]; af[ai].lo = Wl[1024 + fr];
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int off = (64 * sg + 32 * bi + j) * X3ROW + 16 * ks + 8 * hf;
      bf[bi].hi = *reinterpret_cast<const f16x8*>(Xl + off);
      bf[bi].lo = *reinterpret_cast<const f16x8*>(Xl + FX_MB * X3ROW + off);
    }
#pragma unroll
    for (int ai = 0; ai < 2; ++ai)
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) acc[ai][bi] = mfma_x2(af[ai], bf[bi], acc[ai][bi]);
  }
}

// Forward of one network for one 128-sample tile: x → h1 (recomputed slab by slab, optionally stored) → h2 (stored: the backward pass
// needs it) → head Z. Replaces wide_dense_kernel (layer 1) + wide_dense_x2_kernel<EPI_TANH> (layer 2 + head): ppo.jl:35,213-216.
template <int DP, bool WRITE_H1>
__device__ __forceinline__ void wide_fused_fwd_body(const FusedFwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, sg = wave >> 2;
  const int m0 = blockIdx.x * FX_MB;
  // staging role of this thread: sample sm, k-octet sq of every slab
  const int sm = tid & (FX_MB - 1), sq = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int gm = m0 + sm;
  const bool live = gm < a.M;
  float x[DP];
  {
    const int src = live ? (a.perm ? a.perm[gm] : gm) : 0;
    const float* xp = a.obs + (size_t)src * a.D;
#pragma unroll
    for (int c = 0; c < DP; ++c) x[c] = (live && c < a.D) ? xp[c] : 0.0f;
  }
  cfloat_k* W1c = (cfloat_k*)(a.W1s);
  cfloat_k* b1c = W1c + 256 * DP;
  // h1 slab s → LDS (split into fp16 pieces, carried as 2^14·h1) [+ HBM]
  auto stage = [&](int s, unsigned char* xbuf) {
    const int u0 = __builtin_amdgcn_readfirstlane(32 * s + 8 * sq);
    float hv[8];
    // the W1 rows of four units at a time: their scalar loads are issued together (one exposed scalar-memory latency per half, not per unit)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float wr[4][DP], bb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bb[e] = b1c[u0 + 4 * half + e];
#pragma unroll
        for (int c = 0; c < DP; ++c) wr[e][c] = W1c[(u0 + 4 * half + e) * DP + c];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = bb[e];
#pragma unroll
        for (int c = 0; c < DP; ++c) t = __builtin_fmaf(wr[e][c], x[c], t);
        hv[4 * half + e] = tanh_exp2_arg(t, X2_ACT_SCALE);
      }
    }
    const P2 p = split2(hv);
    _Float16* Xl = reinterpret_cast<_Float16*>(xbuf);
    *reinterpret_cast<f16x8*>(Xl + sm * X3ROW + 8 * sq) = p.hi;
    *reinterpret_cast<f16x8*>(Xl + FX_MB * X3ROW + sm * X3ROW + 8 * sq) = p.lo;
    if (WRITE_H1 && live) {
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = hv[e] * (1.0f / X2_ACT_SCALE); o1[e] = hv[4 + e] * (1.0f / X2_ACT_SCALE); }
      float* dst = a.H1 + (size_t)256 * gm + u0;
      *reinterpret_cast<f32x4*>(dst) = o0; *reinterpret_cast<f32x4*>(dst + 4) = o1;
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int ai = 0; ai < 2; ++ai)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ai][bi][r] = 0.0f;
  fx_dma_wslab(a.Wx2, 0, smx, wave, lane);
  stage(0, smx + FX_OFF_X);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < 8; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < 8) fx_dma_wslab(a.Wx2, s + 1, smx + nxt * FX_WBYTES, wave, lane);   // the buffer's last readers passed the barrier of slab s - 1
    // the two waves of a SIMD (w and w + 4) take their staging and their MFMA phases in opposite order, so that one's vector work
    // faces the other's matrix work
    if (sg == 0) {
      if (s + 1 < 8) stage(s + 1, smx + FX_OFF_X + nxt * FX_XBYTES);
      fx_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
    } else {
      fx_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
      if (s + 1 < 8) stage(s + 1, smx + FX_OFF_X + nxt * FX_XBYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of slab s + 1 have landed
    __syncthreads();
  }
  // epilogue: h2 = tanh(acc·unscale + b2), head partials, h2 out in whole 128-B lines (tile_tanh_head), heads folded over the 4 row groups
  float* scr = reinterpret_cast<float*>(smx) + wave * (32 * 36);
  const int hs = a.A;
  float* hp_all = reinterpret_cast<float*>(smx + FX_OFF_X);
  float* hp = hp_all + rg * (FX_MB * hs);
  for (int i = lane; i < 64 * hs; i += 64) hp[64 * sg * hs + i] = 0.0f;
  wave_lds_fence();
  const float cs = a.wsc[1] * (1.0f / X2_ACT_SCALE);
#pragma unroll
  for (int ai = 0; ai < 2; ++ai)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
      tile_tanh_head(scr, acc[ai][bi], lane, 64 * rg + 32 * ai, 64 * sg + 32 * bi, m0 + 64 * sg + 32 * bi, a.M, a.b2, a.H2, a.W3t, a.A, hp, hs, cs, true);
  __syncthreads();
  for (int i = tid; i < FX_MB * a.A; i += 512) {
    const int m = i / a.A, aa = i - m * a.A;
    float z = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) z += hp_all[q * (FX_MB * hs) + m * hs + aa];
    if (m0 + m < a.M) a.Z[(size_t)a.ldz * (m0 + m) + aa] = z + a.b3[aa];
  }
}

template <int DP, bool WRITE_H1>
__global__ void __launch_bounds__(512) wide_fused_fwd_kernel(FusedFwdArgs a0, FusedFwdArgs a1) {
  if (blockIdx.y == 0) wide_fused_fwd_body<DP, WRITE_H1>(a0); else wide_fused_fwd_body<DP, WRITE_H1>(a1);
}

// W1·2·log2(e) as [256][DP] rows (zero beyond obs_dim) followed by b1·2·log2(e): what the staging threads read through scalar loads
__global__ void __launch_bounds__(256) wide_pack_w1s_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int H, int D, int DP,
                                                            float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < H * DP) { const int u = i / DP, c = i - u * DP; out[i] = c < D ? W1[u + H * c] * TWO_LOG2E : 0.0f; }
  else if (i < H * DP + H) out[i] = b1[i - H * DP] * TWO_LOG2E;
}

}  // namespace crl

namespace crl {

// ======================================================================================================================================
// Backward of one network, tile-resident: from the stored h2 and the head cotangent to δ2 (stored for the weight-gradient kernel) and
// dW1 / db1 (per-block partials) — replaces wide_skinny_kernel<…, D2> (δ2 formation), wide_dense_x2_kernel<EPI_DTANH> (δ1 = W2ᵀ·δ2 ⊙
// (1 − h1²), ppo.jl:202 pullbacks) and wide_skinny_kernel<8> (dW1 = δ1·xᵀ): the [256 x M] arrays δ1 and h1 are never read or written here.
//   * h2 slabs (32 units x 128 samples, 16 KB of f32) stream from HBM by LDS-DMA, three buffers deep (two slabs in flight);
//   * the staging thread (sample, 8 units) forms δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²), stores it, scales the sample into the fp16 window (per-sample
//     power of two from the bound Σ_a |δ3[a]|·max_k |W3[a][k]|, as wide_dense_x2_kernel does) and leaves the split pieces in LDS;
//   * the 256 x 256 product runs with its operands SWAPPED (mfma(δ2-fragment, W2ᵀ-fragment)): the same registers, the transposed result —
//     lane = hidden unit, registers = samples — so the sums over samples that dW1 / db1 need are per-lane sums over registers;
//   * (1 − h1²) is recomputed from the 8 observation floats of the sample and the lane's own row of W1 (4·r·(1 − r), r = 1/(2^t + 1)).
// Persistent: a block walks tiles blockIdx.x, + gridDim.x, …; its dW1 / db1 sums leave as ONE partial per block.
// ======================================================================================================================================
constexpr int FB_HBYTES = FX_MB * 32 * 4;                    // 16,384: one h2 slab, [sample][32 units] f32
constexpr int FB_OFF_H = FX_OFF_X + 2 * FX_XBYTES;           // 106,496
constexpr int FB_OFF_W3 = FB_OFF_H + 3 * FB_HBYTES;          // 155,648: W3ᵀ rows [A][256] f32
constexpr int FB_AMAX = 8;
static_assert(FB_OFF_W3 + FB_AMAX * 1024 <= 160 * 1024, "fused backward: LDS budget");

struct FusedBwdArgs {
  const float* H2; const float* dZ; int ldd; int A;
  const float* W3t; const float* wmax;
  const float* Wx2b; const float* wsc;
  const float* obs; const int32_t* perm; int D;
  const float* W1s;
  float* D2; float* pW1; float* pB1;
  int M;
};

// one h2 slab (units 32s …, samples m0 …) into LDS as [sample][32 units]: 16 pieces of 1 KB (8 samples x 128 B), two per wave
__device__ __forceinline__ void fb_dma_hslab(const float* H2, int m0, int s, unsigned char* dst, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = i * 8 + wave;                       // samples 8·piece … 8·piece + 7
    const int smp = 8 * piece + (lane >> 3);
    const char* g = reinterpret_cast<const char*>(H2 + (size_t)256 * (m0 + smp) + 32 * s) + (lane & 7) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
  }
}

// acc[bi][ai] += X-slab(samples 64sg + 32bi …)ᵀ-as-A · W-slab(units 64rg + 32ai …)-as-B: result rows = samples, columns = units
__device__ __forceinline__ void fb_compute_slab(const unsigned char* wbuf, const unsigned char* xbuf, int rg, int sg, int lane, f32x16 (&acc)[2][2]) {
  const f16x8* Wl = reinterpret_cast<const f16x8*>(wbuf);
  const _Float16* Xl = reinterpret_cast<const _Float16*>(xbuf);
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    P2 wf[2], xf[2];
#pragma unroll
    for (int ai = 0; ai < 2; ++ai) {
      const int fr = (ks * 8 + 2 * rg + ai) * 64 + lane;
      wf[ai].hi = Wl[fr]; wf[ai].lo = Wl[1024 + fr];
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      const int off = (64 * sg + 32 * bi + j) * X3ROW + 16 * ks + 8 * hf;
      xf[bi].hi = *reinterpret_cast<const f16x8*>(Xl + off);
      xf[bi].lo = *reinterpret_cast<const f16x8*>(Xl + FX_MB * X3ROW + off);
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int ai = 0; ai < 2; ++ai) acc[bi][ai] = mfma_x2(xf[bi], wf[ai], acc[bi][ai]);
  }
}

template <int DP, int NA>   // NA: head outputs kept in registers (a.A <= NA)
__device__ __forceinline__ void wide_fused_bwd_body(const FusedBwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, sg = wave >> 2, j = lane & 31, hf = lane >> 5;
  const int sm = tid >> 2, sq = tid & 3;                       // staging role: sample, unit octet of every slab
  float* W3l = reinterpret_cast<float*>(smx + FB_OFF_W3);
  for (int i = tid; i < a.A * 256; i += 512) W3l[i] = a.W3t[i];   // W3t is [k + 256·a]: rows of one output contiguous
  // epilogue role: this lane's two hidden units and their rows of W1 (×2·log2 e) / b1
  float w1r[2][DP], b1r[2], gW1[2][DP], gB1[2];
#pragma unroll
  for (int ai = 0; ai < 2; ++ai) {
    const int u = 64 * rg + 32 * ai + j;
#pragma unroll
    for (int c = 0; c < DP; ++c) { w1r[ai][c] = a.W1s[u * DP + c]; gW1[ai][c] = 0.0f; }
    b1r[ai] = a.W1s[256 * DP + u]; gB1[ai] = 0.0f;
  }
  const float wunscale = a.wsc[1];
  const int ntiles = a.M / FX_MB;                              // the launcher takes this path only for M % 128 == 0
  __syncthreads();
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = t * FX_MB, gm = m0 + sm;
    // ---- tile set-up: head cotangent of the staging sample, its fp16 scale, its observation quarter
    float dz[NA];
    float bound = 0.0f;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      dz[q] = q < a.A ? a.dZ[(size_t)a.ldd * gm + q] : 0.0f;
      if (q < a.A) bound = __builtin_fmaf(__builtin_fabsf(dz[q]), a.wmax[q], bound);
    }
    float s1, i1;
    pow2_scale(bound, s1, i1);
    float xq[DP / 4];
    {
      const int src = a.perm ? a.perm[gm] : gm;
#pragma unroll
      for (int c = 0; c < DP / 4; ++c) { const int cc = sq * (DP / 4) + c; xq[c] = cc < a.D ? a.obs[(size_t)src * a.D + cc] : 0.0f; }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int ai = 0; ai < 2; ++ai)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bi][ai][r] = 0.0f;
    // δ2 of (sample sm, units 32s + 8sq …) from the h2 slab in LDS: stored, scaled, split, staged
    auto stage = [&](int s, const unsigned char* hbuf, unsigned char* xbuf) {
      const f32x4* hp = reinterpret_cast<const f32x4*>(hbuf + sm * 128 + sq * 32);
      const f32x4 h0 = hp[0], h1v = hp[1];
      float d[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 0.0f;
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        if (q < a.A) {
          const f32x4* wp = reinterpret_cast<const f32x4*>(W3l + q * 256 + 32 * s + 8 * sq);
          const f32x4 w0 = wp[0], w1v = wp[1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { d[e] = __builtin_fmaf(w0[e], dz[q], d[e]); d[4 + e] = __builtin_fmaf(w1v[e], dz[q], d[4 + e]); }
        }
      }
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = d[e] * (1.0f - h0[e] * h0[e]); o1[e] = d[4 + e] * (1.0f - h1v[e] * h1v[e]); }
      float* dst = a.D2 + (size_t)256 * gm + 32 * s + 8 * sq;
      *reinterpret_cast<f32x4*>(dst) = o0; *reinterpret_cast<f32x4*>(dst + 4) = o1;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = o0[e] * s1; v[4 + e] = o1[e] * s1; }
      const P2 p = split2(v);
      _Float16* Xl = reinterpret_cast<_Float16*>(xbuf);
      *reinterpret_cast<f16x8*>(Xl + sm * X3ROW + 8 * sq) = p.hi;
      *reinterpret_cast<f16x8*>(Xl + FX_MB * X3ROW + sm * X3ROW + 8 * sq) = p.lo;
    };
    unsigned char* Hb = smx + FB_OFF_H;
    fb_dma_hslab(a.H2, m0, 0, Hb, wave, lane);
    fb_dma_hslab(a.H2, m0, 1, Hb + FB_HBYTES, wave, lane);
    fx_dma_wslab(a.Wx2b, 0, smx, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(0, Hb, smx + FX_OFF_X);
    fb_dma_hslab(a.H2, m0, 2, Hb + 2 * FB_HBYTES, wave, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      const int h1i = (s + 1) % 3, h3i = s % 3;                 // buffers of the h2 slabs s + 1 (staged now) and s + 3 (requested now)
      if (s < 7) fx_dma_wslab(a.Wx2b, s + 1, smx + nxt * FX_WBYTES, wave, lane);
      asm volatile("" ::: "memory");   // the δ2 stores below stay behind the weight pieces in issue order (the counted wait relies on it)
      if (sg == 0) {
        if (s < 7) stage(s + 1, Hb + h1i * FB_HBYTES, smx + FX_OFF_X + nxt * FX_XBYTES);
        if (s + 3 < 8) fb_dma_hslab(a.H2, m0, s + 3, Hb + h3i * FB_HBYTES, wave, lane);
        fb_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
      } else {
        fb_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
        if (s < 7) stage(s + 1, Hb + h1i * FB_HBYTES, smx + FX_OFF_X + nxt * FX_XBYTES);
        if (s + 3 < 8) fb_dma_hslab(a.H2, m0, s + 3, Hb + h3i * FB_HBYTES, wave, lane);
      }
      // the weight slab s + 1 (and every older transfer, h2 slab s + 2 among them) has landed; the h2 slab s + 3 and the two δ2 stores
      // of this iteration's staging — issued after the weight pieces in both orders — may stay in flight
      if (s < 5) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else if (s < 7) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // ---- epilogue: δ1ᵀ = acc·unscale ⊙ (1 − h1²) with h1 recomputed; dW1 / db1 accumulate per lane (lane = unit, registers = samples)
    float* xs = reinterpret_cast<float*>(smx + FX_OFF_X + FX_XBYTES);      // [128][DP], in the second activation buffer (free until slab 1 of the next tile)
    float* invs = xs + FX_MB * DP;
#pragma unroll
    for (int c = 0; c < DP / 4; ++c) xs[sm * DP + sq * (DP / 4) + c] = xq[c];
    if (sq == 0) invs[sm] = i1 * wunscale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int srow0 = 64 * sg + 32 * bi + 8 * g + 4 * hf;               // four consecutive samples: registers 4g … 4g + 3
        const f32x4 iv = *reinterpret_cast<const f32x4*>(invs + srow0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xv[DP];
#pragma unroll
          for (int c4 = 0; c4 < DP / 4; ++c4) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(xs + (srow0 + e) * DP + 4 * c4);
            xv[4 * c4] = t4[0]; xv[4 * c4 + 1] = t4[1]; xv[4 * c4 + 2] = t4[2]; xv[4 * c4 + 3] = t4[3];
          }
#pragma unroll
          for (int ai = 0; ai < 2; ++ai) {
            float pre = b1r[ai];
#pragma unroll
            for (int c = 0; c < DP; ++c) pre = __builtin_fmaf(w1r[ai][c], xv[c], pre);
            const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(pre) + 1.0f);
            const float sech2 = 4.0f * rr * (1.0f - rr);                    // 1 − tanh², tanh = 1 − 2r
            const float d1 = acc[bi][ai][4 * g + e] * iv[e] * sech2;
            gB1[ai] += d1;
#pragma unroll
            for (int c = 0; c < DP; ++c) gW1[ai][c] = __builtin_fmaf(d1, xv[c], gW1[ai][c]);
          }
        }
      }
    }
    __builtin_amdgcn_s_barrier();   // every wave has read xs / invs before the next tile's set-up may overwrite that region
  }
  // ---- the block's partial: lane halves hold different samples of the same unit, the two sample groups are two waves
  float* red = reinterpret_cast<float*>(smx);                               // [wave][64 units][DP + 1]
#pragma unroll
  for (int ai = 0; ai < 2; ++ai) {
#pragma unroll
    for (int c = 0; c < DP; ++c) gW1[ai][c] += xor32(gW1[ai][c]);
    gB1[ai] += xor32(gB1[ai]);
    if (hf == 0) {
      float* q = red + (wave * 64 + 32 * ai + j) * (DP + 1);
#pragma unroll
      for (int c = 0; c < DP; ++c) q[c] = gW1[ai][c];
      q[DP] = gB1[ai];
    }
  }
  __syncthreads();
  for (int i = tid; i < 256 * (DP + 1); i += 512) {
    const int u = i / (DP + 1), c = i - u * (DP + 1);
    const int w0 = u >> 6, ul = u & 63;                                     // row group of the unit; the two waves w0 and w0 + 4 hold it
    const float v = red[(w0 * 64 + ul) * (DP + 1) + c] + red[((w0 + 4) * 64 + ul) * (DP + 1) + c];
    if (c == DP) a.pB1[(size_t)blockIdx.x * 256 + u] = v;
    else if (c < a.D) a.pW1[(size_t)blockIdx.x * 256 * a.D + u + 256 * c] = v;
  }
}

template <int DP, int NA0>   // network 0 (actor) keeps up to NA0 head cotangents per sample in registers, network 1 (critic) one
__global__ void __launch_bounds__(512) wide_fused_bwd_kernel(FusedBwdArgs a0, FusedBwdArgs a1) {
  if (blockIdx.y == 0) wide_fused_bwd_body<DP, NA0>(a0); else wide_fused_bwd_body<DP, 1>(a1);
}

}  // namespace crl
