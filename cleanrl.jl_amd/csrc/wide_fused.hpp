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

#ifdef CRL_EXP_WSTAMPS
// diagnostic build only (bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide): wall-clock stamps (100 MHz) of the first tile of
// every block of the fused kernels — [kernel 0 fwd / 1 bwd][block][wave][slot], read back by scripts/wstamps_probe.py
__device__ unsigned long long crl_dbg_wstamps[2 * 256 * 8 * 16];
// backward kernel: stamps (low words of the 100 MHz clock) sit in SCALAR registers until the block ends — the LDS-resident version of round 4 cost vector
// registers this kernel does not have and ran 2x slower than production. Six points per build: CRL_BTS_SET 0 = tile level (slots 0, 10, 1, 5, 11, 6),
// 1 = slab 3 in detail (slots 2, 7, 8, 9, 3, 4).
#ifndef CRL_BTS_SET
#define CRL_BTS_SET 0
#endif
constexpr int crl_bts_idx(int slot) {   // (six stamps held across a whole tile spilled 876 bytes per lane: the tile-level points come three per build, sets 0 and 2)
  return CRL_BTS_SET == 0 ? (slot == 0 ? 0 : slot == 10 ? 1 : slot == 1 ? 2 : -1)
       : CRL_BTS_SET == 2 ? (slot == 5 ? 0 : slot == 11 ? 1 : slot == 6 ? 2 : -1)
                          : (slot == 2 ? 0 : slot == 7 ? 1 : slot == 8 ? 2 : slot == 9 ? 3 : slot == 3 ? 4 : slot == 4 ? 5 : -1);
}
#define CRL_BSTAMP(slot) do { if constexpr (crl_bts_idx(slot) >= 0) bts[crl_bts_idx(slot)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#define CRL_WSTAMP_CYC(kern, slot) do { const unsigned bx_ = (kern) ? blockIdx.x : blockIdx.x - 2048u; if ((threadIdx.x & 63) == 0 && bx_ < 256u && blockIdx.y == 0) crl_dbg_wstamps[(((kern) * 256 + bx_) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define CRL_WSTAMP(kern, slot) do { const unsigned bx_ = (kern) ? blockIdx.x : blockIdx.x - 2048u; if ((threadIdx.x & 63) == 0 && bx_ < 256u && blockIdx.y == 0) crl_dbg_wstamps[(((kern) * 256 + bx_) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); } while (0)
#define CRL_GSTAMP(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256u && blockIdx.y == 0) crl_dbg_wstamps[((blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define CRL_GSTAMP(slot) do { } while (0)
#define CRL_WSTAMP(kern, slot) do { } while (0)
#define CRL_BSTAMP(slot) do { } while (0)
#define CRL_WSTAMP_CYC(kern, slot) do { } while (0)
#endif
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
// Every block of the grid streams the SAME 256 KB of weights, and blocks that started together reach the same slab together: fetched in
// the same order, all CUs of an XCD ask one L2 channel for one line at the same moment and the transfer takes ≈2 µs instead of ≈0.5
// (in-kernel stamps, scripts/wstamps_probe.py). `rot` (from the block index) rotates the order of the 32 pieces, so that at any moment
// the CUs are spread over the slab's lines.
// One LDS-DMA piece — 64 lanes x 16 B from (wave-uniform base + per-lane byte offset) to the 1 KB at the wave-uniform LDS address — in the
// SGPR-base + VGPR-offset form, spelled out. __builtin_amdgcn_global_load_lds compiles to a 64-bit VGPR address that is recomputed per
// piece, and each of those vector writes waits until the previous fetch has read the register pair: 8 pieces took the producers of
// wide_fused_fwd_pc_kernel 1.04 µs to ISSUE beside their partner's MFMAs, 0.32 µs in this form (profiles/r04_c3_stamps.txt).
__device__ __forceinline__ void lds_dma16(const void* sbase, unsigned voff, unsigned lds_addr) {
  // M0 = LDS destination; the compiler reserves M0 (it cannot be named as a clobber), so it is saved and restored inside the statement
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// The same with a per-lane 64-bit global address (a gather into LDS: the piece of lane l lands at lds_addr + 16·l resp. + 4·l)
__device__ __forceinline__ void lds_dma16_v(const void* vaddr, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(vaddr), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void lds_dma4_v(const void* vaddr, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(vaddr), "s"(lds_addr) : "memory");
}
// Eight consecutive observation floats (elements k0 … k0 + 7, zero beyond obs_dim) of one gathered row. With obs_dim % 4 == 0 the row is 16-byte
// aligned and the eight floats are TWO 16-byte loads; element by element the same gather is eight instructions of 32 different cache lines each, and
// the CU's address unit takes a cycle per line: in wide_wgrad_gen_kernel, which gathers per 32-sample slab, that was 2.1 of a slab's 5.0 µs
// (in-kernel stamps, profiles/r05_c3_wgrad_stamps.txt).
__device__ __forceinline__ void load_obs8(const float* __restrict__ obs, size_t row, int D, int k0, bool ok, float (&x)[8]) {
  if ((D & 3) == 0) {
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4* p = reinterpret_cast<const f32x4*>(obs + row * (size_t)D + k0);
    const f32x4 a = (ok && k0 < D) ? p[0] : z, b = (ok && k0 + 4 < D) ? p[1] : z;
    x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = (ok && k0 + c < D) ? obs[row * (size_t)D + k0 + c] : 0.0f;
  }
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p; }

__device__ __forceinline__ void fx_dma_wslab(const float* Wx2, int slab, unsigned char* dst, int wave, int lane, int rot = 0) {
  const char* g = reinterpret_cast<const char*>(Wx2) + (size_t)slab * FX_WBYTES;
  const unsigned l0 = lds_addr_of(dst), voff = lane * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = (i * 8 + wave + rot) & 31;
    lds_dma16(g + piece * 1024, voff, l0 + piece * 1024);
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
  const int rot = (int)((blockIdx.x * 5u + (blockIdx.x >> 3)) & 31u);   // blocks b and b + 8 share an XCD: spread both neighbours and XCD mates
  const int m0 = blockIdx.x * FX_MB;
  // staging role of this thread: sample sm, k-octet sq of every slab
  const int sm = tid & (FX_MB - 1), sq = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int gm = m0 + sm;
  const bool live = gm < a.M;
  CRL_WSTAMP(0, 0);
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
    if (WRITE_H1) {   // (the launcher takes this kernel only for M % 128 == 0: every staging sample is live, the store count per slab is fixed)
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
  fx_dma_wslab(a.Wx2, 0, smx, wave, lane, rot);
  stage(0, smx + FX_OFF_X);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  CRL_WSTAMP(0, 1);
#pragma unroll 1
  for (int s = 0; s < 8; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    if (s == 3) CRL_WSTAMP(0, 2);
    if (s == 4) CRL_WSTAMP(0, 4);
    if (s + 1 < 8) fx_dma_wslab(a.Wx2, s + 1, smx + nxt * FX_WBYTES, wave, lane, rot);   // the buffer's last readers passed the barrier of slab s - 1
    asm volatile("" ::: "memory");   // the h1 stores of the staging stay behind the weight pieces in issue order (the counted wait relies on it)
    if (s == 3) CRL_WSTAMP(0, 7);
    // the two waves of a SIMD (w and w + 4) take their staging and their MFMA phases in opposite order, so that one's vector work
    // faces the other's matrix work
    if (sg == 0) {
      if (s + 1 < 8) stage(s + 1, smx + FX_OFF_X + nxt * FX_XBYTES);
      if (s == 3) CRL_WSTAMP(0, 8);
      fx_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
    } else {
      fx_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
      if (s == 3) CRL_WSTAMP(0, 8);
      if (s + 1 < 8) stage(s + 1, smx + FX_OFF_X + nxt * FX_XBYTES);
    }
    if (s == 3) CRL_WSTAMP(0, 3);                        // this wave's own work of slab 3 done (before the wait and the barrier)
    // this wave's DMA pieces of slab s + 1 have landed; the two h1 stores of this slab's staging (issued behind them) may stay in flight:
    // waiting for their acknowledgement too cost 0.6 us per slab (scripts/wstamps_probe.py)
    if (WRITE_H1 && s + 1 < 8) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  CRL_WSTAMP(0, 5);
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
  CRL_WSTAMP(0, 6);
}

template <int DP, bool WRITE_H1>
__global__ void __launch_bounds__(512) wide_fused_fwd_kernel(FusedFwdArgs a0, FusedFwdArgs a1) {
  if (blockIdx.y == 0) wide_fused_fwd_body<DP, WRITE_H1>(a0); else wide_fused_fwd_body<DP, WRITE_H1>(a1);
}

// W1·2·log2(e) as [256][DP] rows (zero beyond obs_dim) followed by b1·2·log2(e): what the staging threads read through scalar loads
__device__ __forceinline__ void wide_pack_w1s_body(int bx, const float* __restrict__ W1, const float* __restrict__ b1, int H, int D, int DP,
                                                   float* __restrict__ out) {
  const int i = bx * 256 + threadIdx.x;
  if (i < H * DP) { const int u = i / DP, c = i - u * DP; out[i] = c < D ? W1[u + H * c] * TWO_LOG2E : 0.0f; }
  else if (i < H * DP + H) out[i] = b1[i - H * DP] * TWO_LOG2E;
}
__global__ void __launch_bounds__(256) wide_pack_w1s_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int H, int D, int DP,
                                                            float* __restrict__ out) {
  wide_pack_w1s_body(blockIdx.x, W1, b1, H, D, DP, out);
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
  // split flavour (D2h != nullptr): δ2 leaves as the very fp16x2 pieces the staging thread makes for its own product — δ2·s1 = hi + lo with the SAMPLE's power
  // of two s1 — in two [M][256] f16 planes (hi at D2h, lo at D2h + 256·M: the bytes of the f32 array), and 1/s1 per sample in d2s; D2 is not written
  _Float16* D2h; float* d2s;
  const float* W1f = nullptr; const float* w1sc = nullptr;   // wide_rs_bwd_kernel: W1 as fp16x2 fragments (pack w1f) and its scale pair, for layer 1 on the matrix pipe
  int nblk = 0;                                              // wide_rs_bwd_kernel: blocks this network's tiles are spread over (0 = the grid's x extent)
  float* pW3 = nullptr;                                      // wide_rs_bwd_kernel: dW3 partials [block][unit·A + a] (the kernel has h2 and δ3 on chip: no sweep over h2)
};

// one h2 slab (units 32s …, samples m0 …) into LDS as [sample][32 units]: 16 pieces of 1 KB (8 samples x 128 B), two per wave
__device__ __forceinline__ void fb_dma_hslab(const float* H2, int m0, int s, unsigned char* dst, int wave, int lane) {
  const unsigned l0 = lds_addr_of(dst), voff = (lane >> 3) * 1024 + (lane & 7) * 16;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = i * 8 + wave;                       // samples 8·piece … 8·piece + 7
    lds_dma16(H2 + (size_t)256 * (m0 + 8 * piece) + 32 * s, voff, l0 + piece * 1024);
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

template <int DP, int NA, bool SPLIT>   // NA: head outputs kept in registers (a.A <= NA); SPLIT: δ2 leaves as f16 planes (FusedBwdArgs::D2h)
#ifndef CRL_ABL_B
#define CRL_ABL_B 0   // timing ablations of the backward kernel (scripts/ablate_bwd.sh): 1 no epilogue arithmetic, 2 no products, 3 no staging, 4 no δ2 store — results are garbage
#endif
__device__ __forceinline__ void wide_fused_bwd_body(const FusedBwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CRL_EXP_WSTAMPS
  unsigned bts[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#endif
  const int rg = wave & 3, sg = wave >> 2, j = lane & 31, hf = lane >> 5;
  const int sm = tid >> 2, sq = tid & 3;                       // staging role: sample, unit octet of every slab
  const int rot = (int)((blockIdx.x * 5u + (blockIdx.x >> 3) + 16u * blockIdx.y) & 31u);
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
  // tile set-up values of the staging role — head cotangent of the sample, its observation quarter — are loaded a tile ahead
  float dz[NA], xq[DP / 4];
  // The observation row of a tile's sample is addressed through the permutation: that index is loaded a TILE ahead (srcn), so that no load of
  // setup_load depends on another one of the same call. Round 4 loaded index and row back to back: the wait between them was an s_waitcnt vmcnt(0)
  // right behind the next tile's first eight LDS-DMA pieces — every tile sat out their HBM round trip before its epilogue (ISA audit, round 5).
  int srcn = 0;
  auto perm_of = [&](int t) { int g = t * FX_MB + sm; g = g < a.M ? g : a.M - 1; return a.perm ? a.perm[g] : g; };
  auto setup_load = [&](int t, float (&dzo)[NA], float (&xo)[DP / 4]) {
    const int gmn = t * FX_MB + sm;
#pragma unroll
    for (int c = 0; c < DP / 4; ++c) { int cc = sq * (DP / 4) + c; cc = cc < a.D ? cc : a.D - 1; xo[c] = a.obs[(size_t)srcn * a.D + cc]; }   // (beyond obs_dim: zeroed below)
#pragma unroll
    for (int q = 0; q < NA; ++q) dzo[q] = q < a.A ? a.dZ[(size_t)a.ldd * gmn + q] : 0.0f;
    srcn = perm_of(t + gridDim.x);
  };
  unsigned char* Hb = smx + FB_OFF_H;
  auto tile_dma = [&](int t) {   // the first transfers of a tile: h2 slabs 0 and 1, weight slab 0
    fb_dma_hslab(a.H2, t * FX_MB, 0, Hb, wave, lane);
    fb_dma_hslab(a.H2, t * FX_MB, 1, Hb + FB_HBYTES, wave, lane);
    fx_dma_wslab(a.Wx2b, 0, smx, wave, lane, rot);
  };
  if ((int)blockIdx.x < ntiles) { srcn = perm_of(blockIdx.x); tile_dma(blockIdx.x); setup_load(blockIdx.x, dz, xq); }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = t * FX_MB, gm = m0 + sm;
    const bool first = t == (int)(blockIdx.x + 8 * gridDim.x);   // (diagnostic builds stamp the block's ninth tile: warm caches)
    if (first) CRL_BSTAMP(0);
    float bound = 0.0f;
#pragma unroll
    for (int q = 0; q < NA; ++q)
      if (q < a.A) bound = __builtin_fmaf(__builtin_fabsf(dz[q]), a.wmax[q], bound);
    float s1, i1;
    pow2_scale(bound, s1, i1);
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
      if constexpr (!SPLIT) {
        float* dst = a.D2 + (size_t)256 * gm + 32 * s + 8 * sq;
        if (CRL_ABL_B != 4) { *reinterpret_cast<f32x4*>(dst) = o0; *reinterpret_cast<f32x4*>(dst + 4) = o1; }
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = o0[e] * s1; v[4 + e] = o1[e] * s1; }
      const P2 p = split2(v);
      if constexpr (SPLIT) {                                   // two 16-byte stores either way: the loop's counted waits do not change
        _Float16* dh = a.D2h + (size_t)256 * gm + 32 * s + 8 * sq;
        *reinterpret_cast<f16x8*>(dh) = p.hi; *reinterpret_cast<f16x8*>(dh + (size_t)256 * a.M) = p.lo;
      }
      _Float16* Xl = reinterpret_cast<_Float16*>(xbuf);
      *reinterpret_cast<f16x8*>(Xl + sm * X3ROW + 8 * sq) = p.hi;
      *reinterpret_cast<f16x8*>(Xl + FX_MB * X3ROW + sm * X3ROW + 8 * sq) = p.lo;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this tile's first transfers (issued before the previous tile's epilogue) have landed
    __builtin_amdgcn_s_barrier();
    if (first) CRL_BSTAMP(10);
    stage(0, Hb, smx + FX_OFF_X);
    fb_dma_hslab(a.H2, m0, 2, Hb + 2 * FB_HBYTES, wave, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (first) CRL_BSTAMP(1);
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (first && s == 3) CRL_BSTAMP(2);
      if (first && s == 4) CRL_BSTAMP(4);
      const int h1i = (s + 1) % 3, h3i = s % 3;                 // buffers of the h2 slabs s + 1 (staged now) and s + 3 (requested now)
      if (s < 7) fx_dma_wslab(a.Wx2b, s + 1, smx + nxt * FX_WBYTES, wave, lane, rot);
      asm volatile("" ::: "memory");   // the δ2 stores below stay behind the weight pieces in issue order (the counted wait relies on it)
      if (first && s == 3) CRL_BSTAMP(7);
      if (sg == 0) {
        if (s < 7 && CRL_ABL_B != 3) stage(s + 1, Hb + h1i * FB_HBYTES, smx + FX_OFF_X + nxt * FX_XBYTES);
        if (first && s == 3) CRL_BSTAMP(8);
        if (s + 3 < 8) fb_dma_hslab(a.H2, m0, s + 3, Hb + h3i * FB_HBYTES, wave, lane);
        if (first && s == 3) CRL_BSTAMP(9);
        if (CRL_ABL_B != 2) fb_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
      } else {
        if (CRL_ABL_B != 2) fb_compute_slab(smx + cur * FX_WBYTES, smx + FX_OFF_X + cur * FX_XBYTES, rg, sg, lane, acc);
        if (first && s == 3) CRL_BSTAMP(8);
        if (s < 7 && CRL_ABL_B != 3) stage(s + 1, Hb + h1i * FB_HBYTES, smx + FX_OFF_X + nxt * FX_XBYTES);
        if (first && s == 3) CRL_BSTAMP(9);
        if (s + 3 < 8) fb_dma_hslab(a.H2, m0, s + 3, Hb + h3i * FB_HBYTES, wave, lane);
      }
      // the weight slab s + 1 (and every older transfer, h2 slab s + 2 among them) has landed; the h2 slab s + 3 and the two δ2 stores
      // of this iteration's staging — issued after the weight pieces in both orders — may stay in flight
      if (first && s == 3) CRL_BSTAMP(3);
      if (s < 5) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else if (s < 7) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (first) CRL_BSTAMP(5);
    // ---- epilogue: δ1ᵀ = acc·unscale ⊙ (1 − h1²) with h1 recomputed; dW1 / db1 accumulate per lane (lane = unit, registers = samples)
    float* xs = reinterpret_cast<float*>(smx + FX_OFF_X + FX_XBYTES);      // [128][DP], in the second activation buffer (free until slab 1 of the next tile)
    float* invs = xs + FX_MB * DP;
#pragma unroll
    for (int c = 0; c < DP / 4; ++c) xs[sm * DP + sq * (DP / 4) + c] = (sq * (DP / 4) + c < a.D) ? xq[c] : 0.0f;
    if (sq == 0) invs[sm] = i1 * wunscale;
    if constexpr (SPLIT) { if (sq == 0) a.d2s[gm] = i1; }
    // the next tile's first transfers and set-up loads go out now: they land under the epilogue's vector work (every buffer they touch —
    // h2 buffers 0 and 1, weight buffer 0 — was last read before the loop's final barrier)
    const int tn = t + gridDim.x;
    if (tn < ntiles) { tile_dma(tn); setup_load(tn, dz, xq); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (first) CRL_BSTAMP(11);
#pragma unroll
    for (int bi = 0; bi < (CRL_ABL_B == 1 ? 0 : 2); ++bi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int srow0 = 64 * sg + 32 * bi + 8 * g + 4 * hf;               // four consecutive samples: registers 4g … 4g + 3
        const f32x4 iv = *reinterpret_cast<const f32x4*>(invs + srow0);
        // eight independent chains at a time (4 samples x 2 units): the scheduler keeps a dependent chain as written, so they are
        // written interleaved — c outermost
        float xv[4][DP], pre[4][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int c4 = 0; c4 < DP / 4; ++c4) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(xs + (srow0 + e) * DP + 4 * c4);
            xv[e][4 * c4] = t4[0]; xv[e][4 * c4 + 1] = t4[1]; xv[e][4 * c4 + 2] = t4[2]; xv[e][4 * c4 + 3] = t4[3];
          }
          pre[e][0] = b1r[0]; pre[e][1] = b1r[1];
        }
#pragma unroll
        for (int c = 0; c < DP; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) { pre[e][0] = __builtin_fmaf(w1r[0][c], xv[e][c], pre[e][0]); pre[e][1] = __builtin_fmaf(w1r[1][c], xv[e][c], pre[e][1]); }
        float d1[4][2];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ai = 0; ai < 2; ++ai) pre[e][ai] = __builtin_amdgcn_exp2f(pre[e][ai]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ai = 0; ai < 2; ++ai) pre[e][ai] = __builtin_amdgcn_rcpf(pre[e][ai] + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ai = 0; ai < 2; ++ai) {
            const float rr = pre[e][ai];
            d1[e][ai] = acc[bi][ai][4 * g + e] * iv[e] * (4.0f * rr * (1.0f - rr));   // 1 − tanh² = 4r(1 − r), tanh = 1 − 2r
            gB1[ai] += d1[e][ai];
          }
#pragma unroll
        for (int c = 0; c < DP; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) { gW1[0][c] = __builtin_fmaf(d1[e][0], xv[e][c], gW1[0][c]); gW1[1][c] = __builtin_fmaf(d1[e][1], xv[e][c], gW1[1][c]); }
      }
    }
    __builtin_amdgcn_s_barrier();   // every wave has read xs / invs before the next tile's set-up may overwrite that region
    if (first) CRL_BSTAMP(6);
  }
#ifdef CRL_EXP_WSTAMPS
  if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 256) {
    constexpr int slots[3][6] = {{0, 10, 1, -1, -1, -1}, {2, 7, 8, 9, 3, 4}, {5, 11, 6, -1, -1, -1}};
#pragma unroll
    for (int q = 0; q < 6; ++q) if (slots[CRL_BTS_SET][q] >= 0) crl_dbg_wstamps[((256 + blockIdx.x) * 8 + wave) * 16 + slots[CRL_BTS_SET][q]] = bts[q];
  }
#endif
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

template <int DP, int NA0, bool SPLIT>   // network 0 (actor) keeps up to NA0 head cotangents per sample in registers, network 1 (critic) one
__global__ void __launch_bounds__(512) wide_fused_bwd_kernel(FusedBwdArgs a0, FusedBwdArgs a1) {
  if (blockIdx.y == 0) wide_fused_bwd_body<DP, NA0, SPLIT>(a0); else wide_fused_bwd_body<DP, 1, SPLIT>(a1);
}

}  // namespace crl

#ifdef CRL_EXP_WSTAMPS
extern "C" int32_t crl_debug_read_wstamps(unsigned long long* out, int32_t n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(crl::crl_dbg_wstamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif

namespace crl {

// ======================================================================================================================================
// Forward, producer / consumer form (the default; wide_fused_fwd_kernel above is kept as option wide_fuse_pc = 0).
// In-kernel stamps of the symmetric kernel (scripts/wstamps_probe.py) showed why it stays at ~20 % of the matrix pipe: every wave stages
// AND multiplies inside one barrier interval, so a slab costs the SUM of the two latencies (0.8 + 0.6 µs), although the weight stream
// itself is fast (scripts/micro/wstream_rate.hip: 0.35 µs per 32 KB slab). Here the block's waves take fixed roles:
//   waves 0-3  consumers: nothing but fragment reads and MFMAs — wave c owns rows 64c … 64c + 63 x all 128 samples (2 x 4 tiles, 128
//              accumulator registers): 12 fragment reads per 24 MFMAs;
//   waves 4-7  producers (the SIMD partners of the consumers): the weight slab's LDS-DMA and the next activation slab. Layer 1 runs on
//              the matrix pipe as well: h1pre(32 units x 32 samples) = W1-fragment · x-fragment, one fp16x2 product (K = obs_dim <= 16 fits
//              ONE k-step; x scaled per sample, W1 per network into the fp16 window), then tanh, split, 8-byte LDS stores — no scalar loads,
//              ~150 vector instructions per slab under the consumers' 48 MFMAs;
// so a slab costs max(producer, consumer). Persistent: a block walks tiles blockIdx.x, + gridDim.x, …; the producers prepare the next
// tile's first slab and prefetch its observations while the consumers run the epilogue (tanh, head partials, h2 out in whole lines).
// ======================================================================================================================================
// LDS layout, by the number of weight buffers NWB: [NWB weight slabs 32 KB][2 activation slabs 20 KB][W1 fragments 16 KB][b1 1 KB][W3ᵀ][b2 1 KB].
// NWB = 3 (round 5, option wide_fwd_wbufs = 3, n_act <= 6): the CONSUMERS fetch the weight slab after next behind their own MFMAs — two slabs of lead
// for the LDS-DMA, nothing but h1 left to the producers. Built on the reading that the loop (1.44 µs per slab against 0.77 µs of MFMAs) waits for the
// DMA's latency; the stamps of both flavours (profiles/r05_c3_fwd_stamps.txt) say otherwise: with two buffers a producer spends 0.56 µs issuing its 8
// pieces and 0.88 µs on h1 beside a consumer whose 48 MFMAs take 1.04 µs; with three the consumer's 48 MFMAs + 8 pieces take 1.28 µs and the producer's h1
// work 1.40 µs — the same 1.5 µs per slab. Both waves of a SIMD draw on ONE vector-issue port: an MFMA holds it 8 of its 32 cycles, a VALU instruction 4-8,
// and an LDS-DMA piece 100-140 cycles (MI355X_MICROARCH.md: 60 among bare MFMAs, 100-185 in a busy phase) — a slab's 32 pieces cost each SIMD ≈ 0.4 µs of
// issue time whoever issues them. Kept as an option; NWB = 2 stays the default.
// NWB = 0 (round 5, option wide_fwd_wbufs = 0): NO weight slab in LDS. In this kernel a consumer is the only reader of its 64 rows of the weight slab, so
// staging them in LDS buys no sharing: each consumer loads its eight A-fragments of the next slab (8 x global_load_dwordx4, 32 VGPRs) straight from L2 while it
// multiplies the current one; the producers only make h1. The epilogue scratch gets 18 KB of its own in place of the weight buffers.
constexpr int pc_off_x(int nwb) { return nwb == 0 ? 4 * 32 * 36 * 4 : nwb * FX_WBYTES; }
constexpr int pc_off_w1f(int nwb) { return pc_off_x(nwb) + 2 * FX_XBYTES; }
constexpr int pc_off_b1(int nwb) { return pc_off_w1f(nwb) + 16384; }
constexpr int pc_off_w3(int nwb) { return pc_off_b1(nwb) + 1024; }
constexpr int pc_amax(int nwb) { return nwb == 3 ? 6 : 8; }                  // head rows the W3ᵀ table holds (160 KB of LDS: 3 buffers leave room for 6)
constexpr int pc_off_b2(int nwb) { return pc_off_w3(nwb) + pc_amax(nwb) * 1024; }
constexpr int pc_lds(int nwb) { return pc_off_b2(nwb) + 1024; }
static_assert(pc_lds(3) <= 160 * 1024 && pc_lds(2) == 133120, "producer / consumer forward: LDS budget");
constexpr int PC_OFF_W1F = FX_OFF_X + 2 * FX_XBYTES;          // (two-buffer layout) 106,496: W1 A-fragments, [slab 8][piece 2][lane 64][8 halves] = 16 KB
constexpr int PC_OFF_B1 = PC_OFF_W1F + 16384;                // b1·2·log2(e) [256] f32
#ifndef CRL_ABL_F
#define CRL_ABL_F 0   // timing ablations of the forward's epilogue (results are garbage): 1 no tanh, 2 no head partials, 3 no h2 stores
#endif
#ifndef CRL_FWD_DIRECT_H2
#define CRL_FWD_DIRECT_H2 1   // 1: the consumers store h2 straight from their accumulators (default: 570 vs 583 µs per launch); 0: through the LDS transposition (whole 128-B lines per instruction)
#endif
#ifndef PC_PRODUCER_PRIO
#define PC_PRODUCER_PRIO 0   // measured: 0, 2 and 3 within noise (45.5-46.2 ms per C3 iteration on one box)
#endif
constexpr int PC_AMAX = 8;
constexpr int PC_OFF_W3 = PC_OFF_B1 + 1024;                  // W3ᵀ [A <= 8][256] f32 and b2 [256] f32: the consumers' epilogue reads them from LDS
constexpr int PC_OFF_B2 = PC_OFF_W3 + PC_AMAX * 1024;
constexpr int PC_LDS = PC_OFF_B2 + 1024;                     // 133,120 bytes
static_assert(4 * 32 * 36 * 4 <= FX_WBYTES && 4 * FX_MB * PC_AMAX * 4 <= FX_XBYTES, "epilogue scratch aliases weight buffer 1, the head partials activation buffer 1");

struct FusedFwdPCArgs {
  const float* obs; const int32_t* perm; int D;
  const float* W1f;                               // fp16x2 A-fragments of W1·2·log2(e)·scale1 (pack: w1f), then b1·2·log2(e) [256] f32
  const float* w1sc;                              // {scale1, 1/scale1}
  const float* Wx2; const float* b2; const float* wsc;
  const float* W3t; const float* b3; int A; int ldz;
  float* H1; float* H2; float* Z; int M;
  int nblk = 0;                                   // wide_rs_fwd_kernel: blocks this network's tiles are spread over (0 = the grid's x extent)
};

template <int DP, bool WRITE_H1, int NWB>
__device__ __forceinline__ void wide_fused_fwd_pc_body(const FusedFwdPCArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  constexpr int OFF_X = pc_off_x(NWB), OFF_W1F = pc_off_w1f(NWB), OFF_B1 = pc_off_b1(NWB), OFF_W3 = pc_off_w3(NWB), OFF_B2 = pc_off_b2(NWB);
  constexpr int SCRB = NWB == 3 ? 2 : NWB == 2 ? 1 : 0;   // weight buffer the consumers' epilogue scratch aliases: free from the loop's last barrier on (3 buffers: last read in slab 5); NWB = 0: the region in front of the activation slabs
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const int ntiles = a.M / FX_MB;
  // W1 fragments + bias table into LDS, once
  for (int i = tid; i < (16384 + 1024) / 16; i += 512)
    reinterpret_cast<f32x4*>(smx + OFF_W1F)[i] = reinterpret_cast<const f32x4*>(a.W1f)[i];
  for (int i = tid; i < a.A * 256; i += 512) reinterpret_cast<float*>(smx + OFF_W3)[i] = a.W3t[i];
  if (tid < 256) reinterpret_cast<float*>(smx + OFF_B2)[tid] = a.b2[tid];
  __syncthreads();
  if (wave >= 4) {
    if (PC_PRODUCER_PRIO) __builtin_amdgcn_s_setprio(PC_PRODUCER_PRIO);   // few instructions, all on the slab's critical path: they go first
    // ------------------------------------------------------------------------------------------------ producer p: sample tile p
    const int p = wave - 4;
    const float w1un = a.w1sc[1];
    f16x8 xhi, xlo; float xinv;
    // this lane's sample of tile t: its observation half (k = 8hf …). The row index comes through the permutation and is loaded a tile AHEAD (srcn):
    // index and row back to back put an s_waitcnt vmcnt(0) — a memory round trip — at the top of every tile's first slab
    int srcn = 0;
    auto perm_of = [&](int t) { int g = t * FX_MB + 32 * p + j; g = g < a.M ? g : a.M - 1; return a.perm ? a.perm[g] : g; };
    const int k0x = (8 * hf < a.D && 8 * hf < DP) ? 8 * hf : 0;
    auto load_x = [&](int t, float (&xr)[8]) {
      if ((a.D & 3) == 0) {
        const f32x4* q = reinterpret_cast<const f32x4*>(a.obs + (size_t)srcn * (size_t)a.D + k0x);
        const f32x4 q0 = q[0], q1 = q[k0x + 4 < a.D ? 1 : 0];
        xr[0] = q0[0]; xr[1] = q0[1]; xr[2] = q0[2]; xr[3] = q0[3]; xr[4] = q1[0]; xr[5] = q1[1]; xr[6] = q1[2]; xr[7] = q1[3];
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) { const int cc = k0x + c < a.D ? k0x + c : a.D - 1; xr[c] = a.obs[(size_t)srcn * (size_t)a.D + cc]; }
      }
      srcn = perm_of(t + gridDim.x);
    };
    auto make_xfrag = [&](float (&xr)[8]) {                             // per-sample power of two into the fp16 window, split
#pragma unroll
      for (int c = 0; c < 8; ++c) if (8 * hf + c >= a.D || 8 * hf >= DP) xr[c] = 0.0f;    // beyond obs_dim (load_x read a clamped element there)
      float m = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c) m = __builtin_fmaxf(m, __builtin_fabsf(xr[c]));
      m = __builtin_fmaxf(m, xor32(m));
      float s1, i1;
      pow2_scale(m, s1, i1);
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = xr[c] * s1;
      const P2 q = split2(v);
      xhi = q.hi; xlo = q.lo; xinv = i1 * w1un;
    };
    // Layer 1 of the whole tile — 8 slabs x one fp16x2 product — is issued in ONE burst when the tile's observations are at hand: at that
    // point (the consumers' epilogue of the previous tile) the matrix pipe is idle, and the slab loop no longer waits 0.36-0.40 µs per slab
    // for three chained MFMAs queued behind the consumer's 48 (profiles/r04_c3_stamps.txt). 128 registers the producers have to spare.
    f32x16 hpre[8];
    auto layer1 = [&]() {
      P2 bf; bf.hi = xhi; bf.lo = xlo;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f16x8* wf = reinterpret_cast<const f16x8*>(smx + OFF_W1F) + (s * 2) * 64 + lane;
        P2 af; af.hi = wf[0]; af.lo = wf[64];
#pragma unroll
        for (int r = 0; r < 16; ++r) hpre[s][r] = 0.0f;
        hpre[s] = mfma_x2(af, bf, hpre[s]);
      }
    };
    auto dma_w = [&](int s, unsigned char* wbuf) {                     // weight slab s: this producer's 8 of the 32 pieces
      const char* g = reinterpret_cast<const char*>(a.Wx2) + (size_t)s * FX_WBYTES + p * 1024;
      const unsigned lds0 = lds_addr_of(wbuf) + p * 1024, voff = lane * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) lds_dma16(g + i * 4096, voff, lds0 + i * 4096);
    };
    auto produce = [&](int t, int s, const f32x16& c, unsigned char* wbuf, unsigned char* xbuf, bool with_dma, bool stamp = false) {
      // with_dma: the producers fetch the weight slab as well — always with two weight buffers; with three only a tile's first two slabs (under
      // the consumers' epilogue): inside the slab loop the consumers fetch the slab after next themselves, in the shadow of their own MFMAs
      if (with_dma && NWB != 0) dma_w(s, wbuf);
      asm volatile("" ::: "memory");
      if (stamp) CRL_WSTAMP(1, 6);
      // h1 slab: units 32s …, this producer's 32 samples
      const float* b1l = reinterpret_cast<const float*>(smx + OFF_B1) + 32 * s + 4 * hf;
      _Float16* Xl = reinterpret_cast<_Float16*>(xbuf);
      const int gm = t * FX_MB + 32 * p + j;
      // sixteen independent chains, written stage by stage: per group of four (bias load, tanh, split, store) the compiler keeps the groups
      // in order — it cannot tell the LDS stores from the next group's bias load — and a lone wave beside a multiplying partner then waits out
      // every link of a 9-deep chain that is only 4 wide
      f32x4 bv[4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) bv[q4] = *reinterpret_cast<const f32x4*>(b1l + 8 * q4);   // registers 4q4 … 4q4 + 3 = units 32s + 8q4 + 4hf + {0..3}
      float tt[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = __builtin_fmaf(c[i], xinv, bv[i >> 2][i & 3]);
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = __builtin_amdgcn_exp2f(tt[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = __builtin_amdgcn_rcpf(tt[i] + 1.0f);
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = __builtin_fmaf(-2.0f * X2_ACT_SCALE, tt[i], X2_ACT_SCALE);   // tanh_exp2_arg, stage by stage
      uint2 hh[4], ll[4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 hv; hv[0] = tt[4 * q4]; hv[1] = tt[4 * q4 + 1]; hv[2] = tt[4 * q4 + 2]; hv[3] = tt[4 * q4 + 3];
        split2x4(hv, 1.0f, hh[q4], ll[q4]);
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        *reinterpret_cast<uint2*>(Xl + (32 * p + j) * X3ROW + 8 * q4 + 4 * hf) = hh[q4];
        *reinterpret_cast<uint2*>(Xl + FX_MB * X3ROW + (32 * p + j) * X3ROW + 8 * q4 + 4 * hf) = ll[q4];
      }
      if (WRITE_H1) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = tt[4 * q4 + e] * (1.0f / X2_ACT_SCALE);
          *reinterpret_cast<f32x4*>(a.H1 + (size_t)256 * gm + 32 * s + 8 * q4 + 4 * hf) = o;
        }
      }
    };
    float xr[8];
    if ((int)blockIdx.x < ntiles) { srcn = perm_of(blockIdx.x); load_x(blockIdx.x, xr); make_xfrag(xr); layer1(); }
    if ((int)blockIdx.x < ntiles) {
      produce(blockIdx.x, 0, hpre[0], smx, smx + OFF_X, true);
      if (NWB == 3) dma_w(1, smx + FX_WBYTES);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int tn = t + gridDim.x;
      __builtin_amdgcn_s_barrier();                                      // B_start: slab 0 is ready, the consumers are done with the previous tile
      const bool st_ = t == (int)(blockIdx.x + 8 * gridDim.x);
      if (st_) CRL_WSTAMP(1, 0);
      if (tn < ntiles) load_x(tn, xr);                                   // the next tile's observations: in flight under the whole tile
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (st_) CRL_WSTAMP(1, 8 + s);
        if (s < 7) {
          produce(t, s + 1, hpre[(s + 1) & 7], smx + ((s + 1) & 1) * FX_WBYTES, smx + OFF_X + ((s + 1) & 1) * FX_XBYTES, NWB == 2, st_ && s == 3);
          if (st_ && s == 3) CRL_WSTAMP(1, 4);
          if (NWB == 3 || NWB == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS stores are done (h1 stores to HBM, if any, may stay in flight)
          // two buffers: the 8 weight pieces have landed (the 4 h1 stores issued behind them may stay in flight)
          else if (WRITE_H1) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
          if (st_ && s == 3) CRL_WSTAMP(1, 7);
        }
        __builtin_amdgcn_s_barrier();
      }
      if (st_) CRL_WSTAMP(1, 1);
      if (tn < ntiles) {                                                 // the next tile's layer 1 and first slab, under the consumers' epilogue (buffer 0: last read in slab 6)
        make_xfrag(xr);
        layer1();
        produce(tn, 0, hpre[0], smx, smx + OFF_X, true);
        if (NWB == 3) dma_w(1, smx + FX_WBYTES);                         // buffer 1: last read in slab 7, before the loop's last barrier
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (st_) CRL_WSTAMP(1, 2);
      __builtin_amdgcn_s_barrier();                                      // B_epi: the consumers' head partials are complete
      if (st_) CRL_WSTAMP(1, 3);
    }
  } else {
    // ------------------------------------------------------------------------------------------------ consumer c: rows 64c … 64c + 63
    const int c = wave;
    const float cs = a.wsc[1] * (1.0f / X2_ACT_SCALE);
    const int hs = a.A;
    float* scr = reinterpret_cast<float*>(smx + SCRB * FX_WBYTES) + c * (32 * 36);        // in weight buffer SCRB (CRL_FWD_DIRECT_H2 = 0 only)
    (void)scr;
    float* hp_all = reinterpret_cast<float*>(smx + OFF_X + FX_XBYTES);                 // in activation buffer 1 (both are free from the loop's last barrier to slab 1 of the next tile)
    float* hp = hp_all + c * (FX_MB * hs);
    // NWB = 0: weight fragments from global memory, HALF a slab (one k-step: 24 MFMAs) ahead in registers, [ai][piece]: two sets of 16 VGPRs that
    // alternate (a whole slab ahead — 64 VGPRs — spilled). The pack buffer holds a slab as [piece][k-step][row tile][lane][8 halves] (FX_WBYTES per slab).
    const f16x8* wg0 = reinterpret_cast<const f16x8*>(a.Wx2) + (2 * c) * 64 + lane;
    auto wload = [&](int s, int ks, f16x8 (&w)[2][2]) {
      const f16x8* g = wg0 + (size_t)s * (FX_WBYTES / 16) + ks * 512;
#pragma unroll
      for (int ai = 0; ai < 2; ++ai) { w[ai][0] = g[ai * 64]; w[ai][1] = g[1024 + ai * 64]; }
    };
    auto kstep = [&](int s, int ks, const f16x8 (&w)[2][2], f32x16 (&acc)[2][4]) {
      const _Float16* Xl = reinterpret_cast<const _Float16*>(smx + OFF_X + (s & 1) * FX_XBYTES);
      P2 bf[4];
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int off = (32 * bi + j) * X3ROW + 16 * ks + 8 * hf;
        bf[bi].hi = *reinterpret_cast<const f16x8*>(Xl + off);
        bf[bi].lo = *reinterpret_cast<const f16x8*>(Xl + FX_MB * X3ROW + off);
      }
#pragma unroll
      for (int ai = 0; ai < 2; ++ai) {
        P2 af; af.hi = w[ai][0]; af.lo = w[ai][1];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) acc[ai][bi] = mfma_x2(af, bf[bi], acc[ai][bi]);
      }
    };
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int m0 = t * FX_MB;
      f16x8 wa[2][2];
      if (NWB == 0) wload(0, 0, wa);                                     // in flight across the barrier
      f32x16 acc[2][4];
#pragma unroll
      for (int ai = 0; ai < 2; ++ai)
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ai][bi][r] = 0.0f;
      __builtin_amdgcn_s_barrier();                                      // B_start
      const bool st_ = t == (int)(blockIdx.x + 8 * gridDim.x);
      if (st_) { CRL_WSTAMP(1, 0); CRL_WSTAMP_CYC(1, 5); }
      if constexpr (NWB == 0) {
        f16x8 wb2[2][2];
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
          if (st_) CRL_WSTAMP(1, 8 + s);
          wload(s, 1, wb2);
          __builtin_amdgcn_sched_barrier(0);
          kstep(s, 0, wa, acc);
          if (s + 1 < 8) wload(s + 1, 0, wa);
          __builtin_amdgcn_sched_barrier(0);
          kstep(s, 1, wb2, acc);
          if (st_ && s == 3) CRL_WSTAMP(1, 4);
          __builtin_amdgcn_s_barrier();
        }
      } else
#pragma unroll 1
      for (int s = 0; s < 8; ++s) {
        if (st_) CRL_WSTAMP(1, 8 + s);
        const int wb = NWB == 3 ? s % 3 : (s & 1);
        const f16x8* Wl = reinterpret_cast<const f16x8*>(smx + wb * FX_WBYTES);
        const _Float16* Xl = reinterpret_cast<const _Float16*>(smx + OFF_X + (s & 1) * FX_XBYTES);
        // three buffers: the weight slab AFTER NEXT — this consumer's 8 of its 32 pieces, one behind each of the slab's first eight products
        // (buffer (s + 2) % 3 was last read in slab s - 1, before the previous barrier)
        const bool dma = NWB == 3 && s + 2 < 8;
        const char* wg = reinterpret_cast<const char*>(a.Wx2) + (size_t)(s + 2) * FX_WBYTES + c * 1024;
        const unsigned wl0 = lds_addr_of(smx + ((s + 2) % 3) * FX_WBYTES) + c * 1024, wvo = lane * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          P2 af[2], bf[4];
#pragma unroll
          for (int ai = 0; ai < 2; ++ai) {
            const int fr = (ks * 8 + 2 * c + ai) * 64 + lane;
            af[ai].hi = Wl[fr]; af[ai].lo = Wl[1024 + fr];
          }
#pragma unroll
          for (int bi = 0; bi < 4; ++bi) {
            const int off = (32 * bi + j) * X3ROW + 16 * ks + 8 * hf;
            bf[bi].hi = *reinterpret_cast<const f16x8*>(Xl + off);
            bf[bi].lo = *reinterpret_cast<const f16x8*>(Xl + FX_MB * X3ROW + off);
          }
#pragma unroll
          for (int ai = 0; ai < 2; ++ai)
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
              acc[ai][bi] = mfma_x2(af[ai], bf[bi], acc[ai][bi]);
              if (dma && ks == 0) { const int pc = ai * 4 + bi; lds_dma16(wg + pc * 4096, wvo, wl0 + pc * 4096); }
            }
        }
        if (st_ && s == 3) CRL_WSTAMP(1, 4);
        // the pieces of slab s + 1 (issued a slab ago; a tile's first two slabs come from the producers) have landed before anybody passes the
        // barrier; the eight of slab s + 2 just issued stay in flight
        if (NWB == 3) { if (dma) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (st_ && s == 3) CRL_WSTAMP(1, 7);
        __builtin_amdgcn_s_barrier();
      }
      if (st_) { CRL_WSTAMP(1, 1); CRL_WSTAMP_CYC(1, 6); }
      // epilogue: h2 = tanh(acc·unscale + b2) out in whole lines, head partials of this consumer's 64 rows. b2 and W3 come from LDS and a
      // row group's 16 head weights per action are read ONCE for its four sample tiles (the first version read them per tile from global
      // memory: 32 dependent round trips, 11.8 of the tile's 24 µs — profiles/r04_c3_stamps.txt)
      float hacc[4][pc_amax(NWB)];
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int aa = 0; aa < pc_amax(NWB); ++aa) hacc[bi][aa] = 0.0f;
#pragma unroll
      for (int ai = 0; ai < 2; ++ai) {
        const int n0 = 64 * c + 32 * ai;
        const float* b2l = reinterpret_cast<const float*>(smx + OFF_B2) + n0 + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b2l + 8 * g);
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ai][bi][4 * g + e] = CRL_ABL_F == 1 ? __builtin_fmaf(acc[ai][bi][4 * g + e], cs, bv[e]) : tanh_exp2(__builtin_fmaf(acc[ai][bi][4 * g + e], cs, bv[e]), TWO_LOG2E, 1.0f);
        }
#if CRL_ABL_F == 3
        if (a.M < 0)   // (timing ablation: no h2 stores)
#endif
#if CRL_FWD_DIRECT_H2
        if (a.H2) {        // (null: values only — the rollout's batched critic pass)
        // h2 straight from the accumulator layout (lane = sample, 4 consecutive units per register quad): 16-byte stores, 32 B per sample and instruction; the
        // eight instructions of a sample's 256 B of this consumer follow each other, so L2 still writes whole lines
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
          float* hrow = a.H2 + (size_t)256 * (m0 + 32 * bi + j) + n0 + 4 * hf;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o; o[0] = acc[ai][bi][4 * g]; o[1] = acc[ai][bi][4 * g + 1]; o[2] = acc[ai][bi][4 * g + 2]; o[3] = acc[ai][bi][4 * g + 3];
            *reinterpret_cast<f32x4*>(hrow + 8 * g) = o;
          }
        }
        }
#else
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) tile_out<EPI_STORE>(scr, acc[ai][bi], lane, n0, m0 + 32 * bi, a.M, nullptr, nullptr, 0, a.H2, 256);
#endif
#pragma unroll
        for (int aa = 0; aa < (CRL_ABL_F == 2 ? 0 : pc_amax(NWB)); ++aa) {
          if (aa < a.A) {
            const float* w3l = reinterpret_cast<const float*>(smx + OFF_W3) + 256 * aa + n0 + 4 * hf;
            f32x4 w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] = *reinterpret_cast<const f32x4*>(w3l + 8 * g);
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
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
      for (int aa = 0; aa < pc_amax(NWB); ++aa) {
        if (aa < a.A) {
#pragma unroll
          for (int bi = 0; bi < 4; ++bi) {
            float pp = hacc[bi][aa];
            pp += xor32(pp);
            if (hf == 0) hp[(32 * bi + j) * hs + aa] = pp;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (st_) CRL_WSTAMP(1, 2);
      __builtin_amdgcn_s_barrier();                                      // B_epi
      if (st_) CRL_WSTAMP(1, 3);
      for (int i = tid; i < FX_MB * a.A; i += 256) {                     // the four consumers fold the partials in fixed order
        const int m = i / a.A, aa = i - m * a.A;
        float z = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) z += hp_all[q * (FX_MB * hs) + m * hs + aa];
        a.Z[(size_t)a.ldz * (m0 + m) + aa] = z + a.b3[aa];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

template <int DP, bool WRITE_H1, int NWB>
__global__ void __launch_bounds__(512) wide_fused_fwd_pc_kernel(FusedFwdPCArgs a0, FusedFwdPCArgs a1) {
  if (blockIdx.y == 0) wide_fused_fwd_pc_body<DP, WRITE_H1, NWB>(a0); else wide_fused_fwd_pc_body<DP, WRITE_H1, NWB>(a1);
}

// fp16x2 A-fragments of W1·2·log2(e)·scale1 for the producers' layer-1 product — [slab][piece][lane][8]: element e of lane l of slab s is
// W1[32s + (l & 31)][8 (l >> 5) + e] (zero beyond obs_dim) — followed by b1·2·log2(e) [256] as f32. scale1 (a power of two, w1sc[0]) puts
// the largest |W1·2·log2 e| into [2^14, 2^15).
__device__ __forceinline__ void wide_pack_w1f_body(int bx, const float* __restrict__ W1, const float* __restrict__ b1, int D, const float* __restrict__ w1sc,
                                                            float* __restrict__ out) {
  const int t = bx * 256 + threadIdx.x;          // (slab, lane)
  if (t < 8 * 64) {
    const int lane = t & 63, s = t >> 6;
    const int u = 32 * s + (lane & 31), k0 = 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (k0 + e) < D ? W1[u + 256 * (k0 + e)] * TWO_LOG2E * w1sc[0] : 0.0f;
    const P2 p2 = split2(x);
    f16x8* dst = reinterpret_cast<f16x8*>(out) + (s * 2) * 64 + lane;
    dst[0] = p2.hi; dst[64] = p2.lo;
  }
  if (t < 256) out[4096 + t] = b1[t] * TWO_LOG2E;        // 16 KB of fragments = 4096 floats, then the bias table
}
__global__ void __launch_bounds__(256) wide_pack_w1f_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D, const float* __restrict__ w1sc,
                                                            float* __restrict__ out) {
  wide_pack_w1f_body(blockIdx.x, W1, b1, D, w1sc, out);
}
// scale1 for wide_pack_w1f_kernel from the largest |W1| (same rule as wide_w2scale_kernel, with the 2·log2(e) factor folded in)
__device__ __forceinline__ void wide_w1scale_body(const float* __restrict__ W1, int n, float* __restrict__ w1sc) {
  __shared__ float sm[4];
  float m = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) m = __builtin_fmaxf(m, __builtin_fabsf(W1[i]) * TWO_LOG2E);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = __builtin_fmaxf(__builtin_fmaxf(sm[0], sm[1]), __builtin_fmaxf(sm[2], sm[3]));
    float s1, i1;
    if (!(m > 0.0f) || !(m < 3.0e38f)) { s1 = 256.0f; i1 = 1.0f / 256.0f; } else pow2_scale(m, s1, i1);
    w1sc[0] = s1; w1sc[1] = i1;
  }
}
__global__ void __launch_bounds__(256) wide_w1scale_kernel(const float* __restrict__ W1, int n, float* __restrict__ w1sc) { wide_w1scale_body(W1, n, w1sc); }

// the two merged repack launches of ensure_pack (wide.hip)
__global__ void __launch_bounds__(256) wide_prep_a_kernel(PrepArgs a) {
  const PrepNet& q = a.n[blockIdx.y];
  const int net = blockIdx.y, bx = blockIdx.x, H = a.H;
  const float* W1 = a.params + q.pbase;
  const float* W2 = W1 + H * a.D + H;
  const float* W3 = W2 + H * H + H;
  float* pk = a.pack + q.kbase;
  if (bx < a.nb_pack) wide_pack_body(bx, a.nb_pack, a.params, a.pack, H, a.D, a.D8, q.NO, q.O8, q.pbase, q.kbase, q.pk);
  else if (bx < a.nb_pack + a.nb_w1s) wide_pack_w1s_body(bx - a.nb_pack, W1, W1 + H * a.D, H, a.D, a.D8, pk + q.pk.w1s);
  else if (bx == a.nb_pack + a.nb_w1s) wide_w2scale_body(W2, H * H, a.wsc + 2 * net);
  else if (bx == a.nb_pack + a.nb_w1s + 1) wide_w1scale_body(W1, H * a.D, a.wsc + 4 + 2 * net);
  else if (threadIdx.x < 64) wide_wmax_body(W3, q.NO, H, pk + q.pk.wmax);
}
__global__ void __launch_bounds__(256) wide_prep_b_kernel(PrepArgs a) {
  const PrepNet& q = a.n[blockIdx.y];
  const int net = blockIdx.y, bx = blockIdx.x, H = a.H;
  const float* W1 = a.params + q.pbase;
  if (bx < 64) wide_pack_x2_body(bx, a.params, a.pack, q.pbase + H * a.D + H, q.kbase, q.pk, a.wsc + 2 * net);
  else wide_pack_w1f_body(bx - 64, W1, W1 + H * a.D, a.D, a.wsc + 4 + 2 * net, a.pack + q.kbase + q.pk.w1f);
}

}  // namespace crl

namespace crl {

// ======================================================================================================================================
// Weight gradient of the hidden layer with h1 REGENERATED on the CU: dW2[n, k] = Σ_m δ2[n, m]·h1[k, m], db2[n] = Σ_m δ2[n, m] over one sample
// chunk (ppo.jl:202 pullback of Dense(256, 256)). wide_wgrad_x2_kernel reads both [256 x M] operands from HBM (δ2 twice: two 128-column
// blocks); here a block owns the whole 256 x 256 tile (2 x 4 MFMA tiles per wave, 128 accumulator registers), reads δ2 ONCE and never reads
// h1: every wave produces one 32-unit tile of the h1 slab per 32 samples with ONE fp16x2 product on the matrix pipe — mfma(x-fragment,
// W1ᵀ-fragment): rows = samples, columns = units, i.e. lane = unit, registers = 4 consecutive samples per quad, exactly the [unit][sample]
// image the K = samples product wants — then tanh, split, 8-byte LDS stores. With it the forward pass no longer stores h1 at all:
// 2.1 GB of HBM traffic per optimiser step less (h1 written once, read once per network).
// ======================================================================================================================================
struct WgradGenArgs {
  const float* dY; const float* obs; const int32_t* perm; int D;
  const float* W1f; const float* w1sc;
  const float* bz; int bld; int bA; const float* wmax;
  float* pW; float* pB; int M; int chunk;
};

template <int DP>
__device__ __forceinline__ void wide_wgrad_gen_body(const WgradGenArgs& a) {
  constexpr int H = 256, NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  _Float16* Yp = reinterpret_cast<_Float16*>(smw);             // [2][256][X3ROW]: δ2·G pieces, [unit][sample]
  _Float16* Xp = Yp + 2 * H * X3ROW;                           // [2][256][X3ROW]: 2^14·h1 pieces, [unit][sample]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, hf = lane >> 5;
  const int wn = wave & 3, wk = wave >> 2;
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;
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
  // this wave's h1 tile: units 32·wave …; its W1ᵀ B-fragment (= the A-fragment of W1 the forward's producers use) and bias stay in registers
  P2 w1b;
  {
    const f16x8* wf = reinterpret_cast<const f16x8*>(a.W1f) + (wave * 2) * 64 + lane;
    w1b.hi = wf[0]; w1b.lo = wf[64];
  }
  const float b1u = a.W1f[4096 + 32 * wave + j];
  const float w1un = a.w1sc[1];
  f32x16 acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  const int sg = lane & 7, ql = lane >> 3;
  const int rrow = 32 * wave + 4 * ql;
  const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 yr[4], bacc = zero4;
  float xo[8];                                                 // observation half (k = 8hf …) of sample m + (lane & 31)
  int src_nx = (c0 + j < c1) ? (a.perm ? a.perm[c0 + j] : c0 + j) : 0;
  // The next slab's loads. NO lane-dependent branch may surround a load here: behind an `if (row < c1) load` the compiler loses count of the loads in
  // flight and falls back to s_waitcnt vmcnt(0) — round 4's fetch had one before its observation loads and one behind its index load, i.e. every slab
  // waited out the HBM round trip of the δ2 loads it had just issued: 2.1 of a slab's 5.0 µs (in-kernel stamps, profiles/r05_c3_wgrad_stamps.txt). So
  // every lane loads from a clamped, valid address and what lies beyond the chunk / beyond obs_dim is zeroed where it is USED (a slab later). Order: the
  // observation rows first — their address is the index the PREVIOUS fetch loaded last, complete long ago —, then δ2, then the next index.
  const int k0x = (8 * hf < a.D && 8 * hf < DP) ? 8 * hf : 0;
  const bool ragged = ((c1 - c0) & 31) != 0;                   // (uniform) the chunk's last slab is partial
  auto fetch_y = [&](int m) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int mm = m + 4 * sg + e; mm = mm < c1 ? mm : c1 - 1;
      yr[e] = *reinterpret_cast<const f32x4*>(a.dY + (size_t)H * mm + rrow);
    }
  };
  auto fetch_x = [&](int m) {
    if ((a.D & 3) == 0) {
      const f32x4* p = reinterpret_cast<const f32x4*>(a.obs + (size_t)src_nx * (size_t)a.D + k0x);
      const f32x4 q0 = p[0], q1 = p[k0x + 4 < a.D ? 1 : 0];
      xo[0] = q0[0]; xo[1] = q0[1]; xo[2] = q0[2]; xo[3] = q0[3]; xo[4] = q1[0]; xo[5] = q1[1]; xo[6] = q1[2]; xo[7] = q1[3];
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) { const int cc = k0x + c < a.D ? k0x + c : a.D - 1; xo[c] = a.obs[(size_t)src_nx * (size_t)a.D + cc]; }
    }
    int mn = m + 32 + j; mn = mn < c1 ? mn : c1 - 1;
    src_nx = a.perm ? a.perm[mn] : mn;
  };
  // The two waves of a SIMD (wave w and w + 4: wk = 0 / 1) request the next slab's δ2 at DIFFERENT points of the slab. Issuing 32 KB of loads takes a CU
  // 0.9 µs (the "loads issued" interval of the timeline: the wave sits in the issue stage until the memory pipeline has taken its requests), and with all
  // eight waves doing that behind barrier 2 the matrix pipe waited for it. Now waves 4-7 ask for their δ2 rows as soon as they have staged the current ones
  // (their stall runs beside their partner's h1 arithmetic, which gets the issue port to itself meanwhile) and go straight to the MFMAs behind barrier 2,
  // while waves 0-3 issue their loads there as before — beside their partner's MFMAs.
  const bool early_y = wk == 1;
  if (c0 < c1) { fetch_x(c0); fetch_y(c0); }
  for (int m = c0; m < c1; m += 32) {
    const bool gst = m == c0 + 32 * 40;          // (diagnostic builds stamp the chunk's 41st slab)
    if (gst) CRL_GSTAMP(0);
    if (m != c0) __syncthreads();
    if (gst) CRL_GSTAMP(1);
    if (ragged) {                                              // rows beyond the chunk were loaded from its last row: they count as zero
#pragma unroll
      for (int e = 0; e < 4; ++e) if (m + 4 * sg + e >= c1) yr[e] = zero4;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) if (8 * hf + c >= a.D || 8 * hf >= DP) xo[c] = 0.0f;   // beyond obs_dim (a sample beyond the chunk has δ2 = 0: its h1 does not matter)
    // δ2 slab → [unit][sample] pieces (4 x 4 register transposes, as wide_wgrad_x2_kernel)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x4 vy;
      vy[0] = yr[0][e]; vy[1] = yr[1][e]; vy[2] = yr[2][e]; vy[3] = yr[3][e];
      uint2 hh, ll;
      split2x4(vy, G, hh, ll);
      *reinterpret_cast<uint2*>(Yp + (0 * H + rrow + e) * X3ROW + 4 * sg) = hh;
      *reinterpret_cast<uint2*>(Yp + (1 * H + rrow + e) * X3ROW + 4 * sg) = ll;
    }
    bacc += (yr[0] + yr[1]) + (yr[2] + yr[3]);
    if (early_y && m + 32 < c1) fetch_y(m + 32);
    if (gst) CRL_GSTAMP(2);
    // h1 tile of this wave: one product on the matrix pipe, the slab's observations scaled by ONE power of two (the scale must not vary
    // along the rows of the result: they are the registers here)
    {
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
      for (int g = 0; g < 4; ++g) {                             // registers 4g … 4g + 3 = samples 8g + 4hf + {0..3} of the slab
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = tanh_exp2_arg(__builtin_fmaf(c16[4 * g + e], un1, b1u), X2_ACT_SCALE);
        uint2 hh, ll;
        split2x4(hv, 1.0f, hh, ll);
        *reinterpret_cast<uint2*>(Xp + (0 * H + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = hh;
        *reinterpret_cast<uint2*>(Xp + (1 * H + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = ll;
      }
    }
    if (gst) CRL_GSTAMP(3);
    __syncthreads();
    if (gst) CRL_GSTAMP(4);
    if (m + 32 < c1) { fetch_x(m + 32); if (!early_y) fetch_y(m + 32); }
    if (gst) CRL_GSTAMP(5);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      P2 af[2], bf[4];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int off = ((wn * 2 + x) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        af[x].hi = *reinterpret_cast<const f16x8*>(Yp + 0 * H * X3ROW + off);
        af[x].lo = *reinterpret_cast<const f16x8*>(Yp + 1 * H * X3ROW + off);
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int off = ((wk * 4 + y) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
        bf[y].hi = *reinterpret_cast<const f16x8*>(Xp + 0 * H * X3ROW + off);
        bf[y].lo = *reinterpret_cast<const f16x8*>(Xp + 1 * H * X3ROW + off);
      }
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = mfma_x2(af[x], bf[y], acc[x][y]);
      if (gst && ks == 0) CRL_GSTAMP(6);
    }
    if (gst) CRL_GSTAMP(7);
  }
  __syncthreads();
  const float un = Ginv * (1.0f / X2_ACT_SCALE);
  float* pw = a.pW + (size_t)blockIdx.x * H * H;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int k = (wk * 4 + y) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (wn * 2 + x) * 32 + 8 * g + 4 * hf;
        f32x4 o; o[0] = acc[x][y][4 * g] * un; o[1] = acc[x][y][4 * g + 1] * un; o[2] = acc[x][y][4 * g + 2] * un; o[3] = acc[x][y][4 * g + 3] * un;
        *reinterpret_cast<f32x4*>(pw + (size_t)H * k + n) = o;
      }
    }
  if (a.pB) {
    float* scr = reinterpret_cast<float*>(smw);
    *reinterpret_cast<f32x4*>(scr + 4 * tid) = bacc;
    __syncthreads();
    if (sg == 0) {
      f32x4 sacc = bacc;
      for (int q = 1; q < 8; ++q) sacc += *reinterpret_cast<const f32x4*>(scr + 4 * (tid + q));
      *reinterpret_cast<f32x4*>(a.pB + (size_t)blockIdx.x * H + rrow) = sacc;
    }
  }
}

template <int DP>
__global__ void __launch_bounds__(512) wide_wgrad_gen_kernel(WgradGenArgs a0, WgradGenArgs a1) {
  if (blockIdx.y == 0) wide_wgrad_gen_body<DP>(a0); else wide_wgrad_gen_body<DP>(a1);
}
constexpr int WG_LDS = 2 * 2 * 256 * X3ROW * 2;   // 81,920 bytes

// ======================================================================================================================================
// The same weight gradient from δ2 as the backward kernel's SPLIT output (FusedBwdArgs::D2h): nothing is converted here, and nothing is loaded into
// registers — every global read of the slab loop is an LDS-DMA, so the loop has no compiler-counted load and its only vmcnt wait is the one written
// below. With the operand staging gone, the loop is a ONE-barrier software pipeline: between two barriers every wave multiplies slab m (48 MFMAs from
// the landed planes and the h1 pieces made during the previous slab) and makes the h1 pieces of slab m + 32 into the other buffer — the matrix work and
// the vector work of a slab no longer alternate in lock-step phases with a barrier and the block's wave skew between them (in-kernel stamps of the
// two-barrier form of this kernel: 1.2 µs of h1 arithmetic with the matrix pipe idle, 1.8 µs of products with the vector pipe idle, 0.9 µs of
// barriers and skew per slab).
// A slab's 32 sample rows of the hi and the lo plane (2 x 16 KB) are requested one slab ahead, between the slab's two jobs (the buffer is free from the
// barrier on; the request costs a wave 0.2 µs of issue); the A operand (rows = units, k = samples) is read from that [sample][unit] image with ds_read_b64_tr_b16 —
// the transposing read — so the image is what the DMA can write: lane-linear rows, 16-byte chunks XORed with the row's low two bits so that the four
// rows of a transposed block sit in four different bank groups (chunk c of row r lies at 512·r + 16·(c ^ ((r & 3) << 2)); the DMA's lanes fetch the
// permuted chunk, the reads undo it). Wave 0 also gathers the observation rows (per-lane addresses) and the inverse scales of the slab after next and
// the permutation indices of the slab after that.
// The planes carry δ2·s1 with a PER-SAMPLE power of two, and the sum runs over samples: the factor 1/s1 goes onto the other operand — the regenerated
// h1 tile is multiplied by t = (1/s1) / max over the chunk of (1/s1), a power of two ≤ 1 (samples with small cotangents lose low bits of h1 exactly as
// they lost low bits of δ2 under the one scale per chunk of wide_wgrad_gen_kernel) — and the chunk's maximum is taken out of the accumulators at the end.
// db2 = Σ δ2 is summed from the A fragments themselves ((hi + lo)·t, exact in f32) by the four waves that hold distinct fragments.
// ======================================================================================================================================
struct WgradSplitArgs {
  const _Float16* Yh; const float* ys;
  const float* obs; const int32_t* perm; int D;
  const float* W1f; const float* w1sc;
  float* pW; float* pB; int M; int chunk;
};
constexpr int WS_YBUF = 2 * 32 * 512;                          // 32,768: one slab, 32 hi rows then 32 lo rows of 512 B
constexpr int WS_XBUF = 2 * 256 * X3ROW * 2;                   // 40,960: [2 pieces][256][X3ROW] f16, the h1 pieces of one slab
constexpr int WS_OFF_X = 2 * WS_YBUF;                          // 65,536
constexpr int WS_OFF_O = WS_OFF_X + 2 * WS_XBUF;               // 147,456: a slab's observation rows as gathered, [2 buffers][2,048 B]
constexpr int WS_OFF_T = WS_OFF_O + 2 * 2048;                  // 151,552: 1/s1 of a slab's samples, [3][64] f32 (32 used)
constexpr int WS_OFF_I = WS_OFF_T + 3 * 256;                   // 152,320: permutation indices of a slab, [2][64] i32 (32 used)
constexpr int WS_LDS = WS_OFF_I + 2 * 256;                     // 152,832
static_assert(WS_LDS <= 160 * 1024, "split weight gradient: LDS budget");
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) h16x4 lds_h16x4;

template <int DP>
__device__ __forceinline__ void wide_wgrad_split_body(const WgradSplitArgs& a) {
  constexpr int H = 256, NT = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, hf = lane >> 5;
  const int wn = wave & 3, wk = wave >> 2;
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = (c0 + a.chunk) < a.M ? (c0 + a.chunk) : a.M;     // (the launcher guarantees whole 32-sample slabs)
  // the chunk's largest 1/s1 (every entry is a power of two: the reciprocal below is exact)
  float imax, iinv;
  {
    float mx = 0.0f;
    for (int m = c0 + tid; m < c1; m += NT) mx = __builtin_fmaxf(mx, a.ys[m]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, o, 64));
    float* red = reinterpret_cast<float*>(smw);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w8 = 1; w8 < NT / 64; ++w8) mx = __builtin_fmaxf(mx, red[w8]);
    __syncthreads();
    imax = mx > 0.0f ? mx : 1.0f;
    iinv = __uint_as_float((254u - ((__float_as_uint(imax) >> 23) & 0xFFu)) << 23);
  }
  P2 w1b;
  {
    const f16x8* wf = reinterpret_cast<const f16x8*>(a.W1f) + (wave * 2) * 64 + lane;
    w1b.hi = wf[0]; w1b.lo = wf[64];
  }
  const float b1u = a.W1f[4096 + 32 * wave + j];
  const float w1un = a.w1sc[1];
  f32x16 acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const int k0x = (8 * hf < a.D && 8 * hf < DP) ? 8 * hf : 0;
  const bool obs16 = (a.D & 3) == 0;
  const unsigned ybase = lds_addr_of(smw);
  // wave 0's side requests (clamped addresses: a lane past the chunk fetches the chunk's last sample, nobody uses it)
  auto gather_obs = [&](int idx, int buf) {      // rows of a slab's samples: lane (j, hf) fetches elements 8hf … of sample j
    const unsigned dst = ybase + WS_OFF_O + buf * 2048;
    if (obs16) {
      const float* p = a.obs + (size_t)idx * (size_t)a.D + k0x;
      lds_dma16_v(p, dst);
      lds_dma16_v(p + (k0x + 4 < a.D ? 4 : 0), dst + 1024);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) { const int cc = k0x + c < a.D ? k0x + c : a.D - 1; lds_dma4_v(a.obs + (size_t)idx * (size_t)a.D + cc, dst + 256 * c); }
    }
  };
  auto fetch_scales = [&](int m, int buf) { int mm = m + lane; mm = mm < c1 ? mm : c1 - 1; lds_dma4_v(a.ys + mm, ybase + WS_OFF_T + buf * 256); };
  auto fetch_index = [&](int m, int buf) { int mm = m + lane; mm = mm < c1 ? mm : c1 - 1; lds_dma4_v(a.perm + mm, ybase + WS_OFF_I + buf * 256); };
  // DMA of one slab's planes: piece p (0..31) = rows 2(p & 15), +1 of plane p >> 4; this wave takes pieces wave, wave + 8, +16, +24 (p & 1 = wave & 1 for all four)
  const unsigned dvoff = (unsigned)(hf * 512 + 16 * (j ^ (((2 * (wave & 1) + hf) & 3) << 2)));
  const _Float16* ylo = a.Yh + (size_t)256 * a.M;
  auto dma_slab = [&](int m, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave + 8 * i;
      const _Float16* plane = (i < 2) ? a.Yh : ylo;
      lds_dma16(plane + (size_t)256 * (m + 2 * (p & 15)), dvoff, ybase + buf * WS_YBUF + (p >> 4) * 16384 + (p & 15) * 1024);
    }
  };
  // transposed-read addresses (bytes from the buffer's base): 16-lane group gq, q = row of the 4-row block, pp = 4-column piece
  const int gq = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int abase = 512 * q4 + 8 * (pp & 1) + 16 * (2 * (gq & 1) + (pp >> 1)) + 4096 * (gq >> 1);
  int aoff[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) aoff[x] = abase + 64 * ((wn * 2 + x) ^ q4);
  auto tr8 = [&](const unsigned char* base, int off) -> f16x8 {   // samples r0 … r0 + 7 of the lane's unit: two 4-row blocks
    const h16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h16x4*)(base + off));
    const h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h16x4*)(base + off + 2048));
    f16x8 r;
    r[0] = (_Float16)u[0]; r[1] = (_Float16)u[1]; r[2] = (_Float16)u[2]; r[3] = (_Float16)u[3];
    r[4] = (_Float16)v[0]; r[5] = (_Float16)v[1]; r[6] = (_Float16)v[2]; r[7] = (_Float16)v[3];
    return r;
  };
  // h1 pieces of one slab, in two halves so that the products of the slab in flight can be issued between them:
  //   pre: the gathered observation rows → one fp16x2 product (rows = samples in registers, columns = units in lanes);
  //   post: tanh, times the samples' t, split, 8-byte stores into the [unit][sample] image
  auto h1_pre = [&](int obuf, float& un1) -> f32x16 {
    float xo[8];
    const float* xs = reinterpret_cast<const float*>(smw + WS_OFF_O + obuf * 2048);
    if (obs16) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(xs + 4 * lane), q1 = *reinterpret_cast<const f32x4*>(xs + 256 + 4 * lane);
      xo[0] = q0[0]; xo[1] = q0[1]; xo[2] = q0[2]; xo[3] = q0[3]; xo[4] = q1[0]; xo[5] = q1[1]; xo[6] = q1[2]; xo[7] = q1[3];
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) xo[c] = xs[64 * c + lane];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) if (8 * hf + c >= a.D || 8 * hf >= DP) xo[c] = 0.0f;
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
    un1 = ix * w1un;
    return mfma_x2(xa, w1b, c16);
  };
  auto h1_post = [&](const f32x16& c16, float un1, int tbuf, int xbuf) {
    const float* tc = reinterpret_cast<const float*>(smw + WS_OFF_T) + 64 * tbuf;
    _Float16* Xp = reinterpret_cast<_Float16*>(smw + WS_OFF_X + xbuf * WS_XBUF);
#pragma unroll
    for (int g = 0; g < 4; ++g) {                               // registers 4g … 4g + 3 = samples 8g + 4hf + {0..3} of the slab
      const f32x4 tq = *reinterpret_cast<const f32x4*>(tc + 8 * g + 4 * hf) * iinv;
      f32x4 hv;
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = tanh_exp2_arg(__builtin_fmaf(c16[4 * g + e], un1, b1u), X2_ACT_SCALE) * tq[e];
      uint2 hh, ll;
      split2x4(hv, 1.0f, hh, ll);
      *reinterpret_cast<uint2*>(Xp + (0 * H + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = hh;
      *reinterpret_cast<uint2*>(Xp + (1 * H + 32 * wave + j) * X3ROW + 8 * g + 4 * hf) = ll;
    }
  };
  // the loop below must not contain a compiler-counted wait: the values loaded above are made "used" here, so their s_waitcnt is emitted before the loop
  // (the compiler does not count the LDS-DMA pieces, so a vmcnt(n) of its own inside the loop would wait for all but n of THEM)
  asm volatile("" :: "v"(w1b.hi), "v"(w1b.lo), "v"(b1u), "v"(w1un));
  if (c0 < c1) {
    dma_slab(c0, 0);
    if (wave == 0) {
      int m0 = c0 + j; m0 = m0 < c1 ? m0 : c1 - 1;
      int m1 = c0 + 32 + j; m1 = m1 < c1 ? m1 : c1 - 1;
      gather_obs(a.perm ? a.perm[m0] : m0, 0);
      gather_obs(a.perm ? a.perm[m1] : m1, 1);
      fetch_scales(c0, 0); fetch_scales(c0 + 32, 1);
      if (a.perm) fetch_index(c0 + 64, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float un1;
    const f32x16 c16 = h1_pre(0, un1);
    h1_post(c16, un1, 0, 0);
  }
  // The two waves of a SIMD (w and w + 4) take a slab's two jobs in OPPOSITE order: a wave issues in order, so its own matrix and vector work never overlap,
  // but its partner's can — while waves 0-3 make their h1 tiles (vector pipe) waves 4-7 run their 48 MFMAs, then they swap. Two copies of the loop, one per
  // order (as an if / else inside one loop body the compiler spilled 400 bytes per lane); both execute the same barriers.
  auto slab_loop = [&](auto h1_first_tag) {
  constexpr bool H1_FIRST = decltype(h1_first_tag)::value;
  int cur = 0, i3 = 0;                                          // parity of the slab, slab index mod 3
  for (int m = c0; m < c1; m += 32, cur ^= 1, i3 = i3 == 2 ? 0 : i3 + 1) {
    const bool more = m + 32 < c1, more2 = m + 64 < c1;
    const bool gst = m == c0 + 32 * 40;          // (diagnostic builds stamp the chunk's 41st slab)
    if (gst) CRL_GSTAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this slab's planes (requested a slab ago) and the side requests of the last slab have landed
    __syncthreads();                                            // … for every wave; the h1 pieces of this slab are complete; the previous slab's products have read their buffers
    if (gst) CRL_GSTAMP(1);
    auto requests = [&]() {
      if (more) dma_slab(m + 32, cur ^ 1);
      if (more2 && wave == 0) {                                 // observation rows and scales of slab m + 64, indices of slab m + 96
        int idx = m + 64 + j;
        if (a.perm) idx = reinterpret_cast<const int*>(smw + WS_OFF_I)[64 * cur + j];
        gather_obs(idx, cur);
        fetch_scales(m + 64, i3 == 0 ? 2 : i3 - 1);
        if (a.perm && m + 96 < c1) fetch_index(m + 96, cur ^ 1);
      }
    };
#if CRL_WS_DMA_AT == 0
    requests();
#endif
    if (gst) CRL_GSTAMP(2);
    const float* tc = reinterpret_cast<const float*>(smw + WS_OFF_T) + 64 * i3;
    const unsigned char* yb = smw + cur * WS_YBUF;
    const _Float16* Xp = reinterpret_cast<const _Float16*>(smw + WS_OFF_X + cur * WS_XBUF);
    auto products = [&]() {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        P2 af[2], bf[4];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          af[x].hi = tr8(yb, aoff[x] + 8192 * ks);
          af[x].lo = tr8(yb + 16384, aoff[x] + 8192 * ks);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          const int off = ((wk * 4 + y) * 32 + j) * X3ROW + 16 * ks + 8 * hf;
          bf[y].hi = *reinterpret_cast<const f16x8*>(Xp + 0 * H * X3ROW + off);
          bf[y].lo = *reinterpret_cast<const f16x8*>(Xp + 1 * H * X3ROW + off);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[x][y] = mfma_x2(af[x], bf[y], acc[x][y]);
        if (wk == 0) {                                          // db2: the fragment's 8 samples of unit 32(2wn + x) + j, scaled back per sample
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tc + 16 * ks + 8 * hf) * iinv, t1 = *reinterpret_cast<const f32x4*>(tc + 16 * ks + 8 * hf + 4) * iinv;
#pragma unroll
          for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              bsum[x] = __builtin_fmaf((float)af[x].hi[e], t0[e], bsum[x]); bsum[x] = __builtin_fmaf((float)af[x].lo[e], t0[e], bsum[x]);
              bsum[x] = __builtin_fmaf((float)af[x].hi[4 + e], t1[e], bsum[x]); bsum[x] = __builtin_fmaf((float)af[x].lo[4 + e], t1[e], bsum[x]);
            }
          }
        }
      }
    };
    auto make_h1 = [&]() {
      if (more) {
        float un1;
        const f32x16 c16 = h1_pre(cur ^ 1, un1);
        h1_post(c16, un1, i3 == 2 ? 0 : i3 + 1, cur ^ 1);
      }
    };
#if CRL_WS_DMA_AT == 0
    if constexpr (H1_FIRST) { make_h1(); if (gst) CRL_GSTAMP(3); products(); }
    else { products(); if (gst) CRL_GSTAMP(3); make_h1(); }
#else
    if constexpr (H1_FIRST) { make_h1(); requests(); if (gst) CRL_GSTAMP(3); products(); }
    else { products(); requests(); if (gst) CRL_GSTAMP(3); make_h1(); }
#endif
    if (gst) CRL_GSTAMP(7);
  }
  };
  if (wk == 0) slab_loop(std::true_type{}); else slab_loop(std::false_type{});
  __syncthreads();
  const float un = imax * (1.0f / X2_ACT_SCALE);
  float* pw = a.pW + (size_t)blockIdx.x * H * H;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int k = (wk * 4 + y) * 32 + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (wn * 2 + x) * 32 + 8 * g + 4 * hf;
        f32x4 o; o[0] = acc[x][y][4 * g] * un; o[1] = acc[x][y][4 * g + 1] * un; o[2] = acc[x][y][4 * g + 2] * un; o[3] = acc[x][y][4 * g + 3] * un;
        *reinterpret_cast<f32x4*>(pw + (size_t)H * k + n) = o;
      }
    }
  if (a.pB && wk == 0) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float v = bsum[x] + xor32(bsum[x]);
      if (hf == 0) a.pB[(size_t)blockIdx.x * H + (wn * 2 + x) * 32 + j] = v * imax;
    }
  }
}

template <int DP>
__global__ void __launch_bounds__(512) wide_wgrad_split_kernel(WgradSplitArgs a0, WgradSplitArgs a1) {
  if (blockIdx.y == 0) wide_wgrad_split_body<DP>(a0); else wide_wgrad_split_body<DP>(a1);
}

}  // namespace crl
