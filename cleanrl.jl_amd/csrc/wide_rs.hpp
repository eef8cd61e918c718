// wide_rs.hpp — register-stationary forward of the 2x256 networks (included by wide.hip behind wide_fused.hpp; BASELINE config C3).
//
// wide_fused_fwd_pc_kernel streams all 256 KB of a network's fp16x2 W2 pieces through LDS for EVERY 128-sample tile — 8 slabs, a barrier and
// 32 LDS-DMA pieces each, 0.4 µs of vector-issue time per slab and SIMD whoever issues them (profiles/r05_c3_fwd_stamps.txt) — and ends at 3.5 x
// its matrix time. But 256 rows x 256 k x two f16 pieces is exactly 8 waves x 128 VGPRs x 64 lanes x 4 B: the block's eight waves can HOLD the
// layer. Here wave w keeps the A-fragments of rows 32w … 32w + 31 (all 16 k-steps, both pieces: 128 registers, loaded once per launch) and the
// only operand that moves is the activation tile:
//   P  wave w makes units 32w … 32w + 31 of h1 for the tile's 32 samples — layer 1 as one fp16x2 product (its W1 fragment comes from LDS), exp2 activation, split, 8-byte LDS stores into a [piece][sample][256 k + 8] tile (33 KB, two of them);
//   M  48 MFMAs per wave: its rows x the tile, B-fragments from LDS (two 16-byte reads per three MFMAs), two accumulators (cross terms,
//      hi·hi), so consecutive MFMAs are independent and the small terms are summed among themselves first;
//   E  h2 = tanh(acc·unscale + b2) out in 16-byte stores (optional), head partials of the wave's 32 rows into LDS; a fold of the 8 partials later.
// No weight traffic, no slab loop, ONE barrier per phase. The two waves of a SIMD (w and w + 4) run half a period apart — waves 0-3 multiply
// tile t while waves 4-7 finish tile t - 1 and prepare their share of t + 1, then the roles swap — so every SIMD always has one wave on the
// matrix pipe and one on the vector pipe:
//   phase 2t    : A = M(t)                               B = E(t-1), P(t+1)
//   phase 2t+1  : A = E(t), P(t+1), fold(t-1)            B = M(t)
// h1(t+1) is written (B in 2t, A in 2t+1) while h1(t) is read (A in 2t, B in 2t+1): two tile buffers; the head partials alternate by tile parity.
#pragma once
#include <type_traits>

namespace crl {

#ifdef CRL_EXP_WSTAMPS
// diagnostic build only (bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide; scripts/rs_stamps.py): [block][wave][slot] wall-clock stamps of two phases
__device__ unsigned long long crl_dbg_rs_stamps[256 * 8 * 16];
#define RS_STAMP(slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256u && blockIdx.y == 0) crl_dbg_rs_stamps[((blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define RS_STAMP(slot) do { } while (0)
#endif
constexpr int RS_MB = 32;                                     // samples per tile
constexpr int RS_ROW = 264;                                   // halves per sample row of an h1 tile: 256 k + 8 pad (528 B: conflict-free b128 reads, b64 writes)
constexpr int RS_XBYTES = 2 * RS_MB * RS_ROW * 2;             // 33,792: one h1 tile, [piece][sample][RS_ROW halves]
constexpr int RS_OFF_B1 = 2 * RS_XBYTES;                      // b1·2·log2(e) [256] f32
constexpr int RS_OFF_W3 = RS_OFF_B1 + 1024;                   // W3ᵀ [A <= 8][256] f32
constexpr int RS_OFF_B2 = RS_OFF_W3 + 8 * 1024;               // b2 [256] f32
constexpr int RS_OFF_HP = RS_OFF_B2 + 1024;                   // head partials [tile parity 2][wave 8][sample 32][8] f32
constexpr int RS_OFF_W1 = RS_OFF_HP + 2 * 8 * RS_MB * 8 * 4;  // W1 fragments [slab 8][piece 2][lane 64] f16x8 (pack w1f): 16 KB
constexpr int RS_LDS = RS_OFF_W1 + 16384;                     // 110,592 bytes

// the stationary operand: this wave's 32 rows of W2 (fp16x2 A-fragments of all 16 k-steps) and of W1 (one k-step)
struct RsWeights { f16x8 wh[16], wl[16]; };
__device__ __forceinline__ void rs_load_weights(const float* Wx2, int wave, int lane, RsWeights& W) {
  const f16x8* wp = reinterpret_cast<const f16x8*>(Wx2) + wave * 64 + lane;       // pack x2f: [slab 8][piece 2][k-step 2][row tile 8][lane 64] f16x8
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      W.wh[2 * s + ks] = wp[s * 2048 + ks * 512];
      W.wl[2 * s + ks] = wp[s * 2048 + 1024 + ks * 512];
    }
}
// this lane's observation half of one sample, scaled per sample into the fp16 window and split (the B-fragment of layer 1)
template <int DP>
__device__ __forceinline__ void rs_xfrag(float (&xr)[8], int D, int hf, P2& xb, float& xinv, float w1un) {
#pragma unroll
  for (int c = 0; c < 8; ++c) if (8 * hf + c >= D || 8 * hf >= DP) xr[c] = 0.0f;
  float m = 0.0f;
#pragma unroll
  for (int c = 0; c < 8; ++c) m = __builtin_fmaxf(m, __builtin_fabsf(xr[c]));
  m = max32(m);
  float s1, i1;
  pow2_scale(m, s1, i1);
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = xr[c] * s1;
  xb = split2(v);
  xinv = i1 * w1un;
}
// layer 1 of the wave's 32 units for a tile: W1 fragment (LDS) x observation fragment, 2·log2(e)·scale folded into the pack — issued at the END of the wave's
// own multiply phase: inside the vector phase the three MFMAs would queue behind the SIMD partner's 48 (0.4 µs of waiting per tile, profiles/r04_c3_stamps.txt)
__device__ __forceinline__ f32x16 rs_layer1(const f16x8* w1l, const P2& xb) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.0f;
  P2 w1; w1.hi = w1l[0]; w1.lo = w1l[64];                                                 // [piece 2][lane 64] f16x8
  return mfma_x2(w1, xb, c);
}
// units 32·wave … of h1 for the tile's 32 samples into the tile buffer, from the layer-1 product c. RAT: the reference's rational tanh_fast (the rollout's
// actor), else the exp2 form
template <bool RAT, int MB = RS_MB>
__device__ __forceinline__ void rs_produce(const f32x16& c, float xinv, const float* b1tab, _Float16* Xl, int wave, int j, int hf) {
  const float* b1l = b1tab + 32 * wave + 4 * hf;
  f32x4 bv[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) bv[q4] = *reinterpret_cast<const f32x4*>(b1l + 8 * q4);   // registers 4q4 … 4q4 + 3 = units 32·wave + 8q4 + 4hf + {0..3}
  float tt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) tt[i] = __builtin_fmaf(c[i], xinv, bv[i >> 2][i & 3]);        // 2·log2(e)·(W1·x + b1)
  if (RAT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[i] = tanh_fast(tt[i] * INV_TWO_LOG2E) * X2_ACT_SCALE;
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[i] = __builtin_amdgcn_exp2f(tt[i]);
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[i] = __builtin_amdgcn_rcpf(tt[i] + 1.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) tt[i] = __builtin_fmaf(-2.0f * X2_ACT_SCALE, tt[i], X2_ACT_SCALE);
  }
  uint2 hh[4], ll[4];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    f32x4 hv; hv[0] = tt[4 * q4]; hv[1] = tt[4 * q4 + 1]; hv[2] = tt[4 * q4 + 2]; hv[3] = tt[4 * q4 + 3];
    split2x4(hv, 1.0f, hh[q4], ll[q4]);
  }
  _Float16* xo = Xl + j * RS_ROW + 32 * wave + 4 * hf;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    *reinterpret_cast<uint2*>(xo + 8 * q4) = hh[q4];
    *reinterpret_cast<uint2*>(xo + MB * RS_ROW + 8 * q4) = ll[q4];
  }
}
// head partials of an ACTIVATED tile (h2 of the wave's 32 rows in aa) into hp [sample 32][8]
template <int NA>
__device__ __forceinline__ void rs_heads(const f32x16& aa, const float* w3tab, float* hp, int wave, int j, int hf) {
  float pp[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    if (a && (a & 1) == 0) __builtin_amdgcn_sched_barrier(0);          // two actions' table reads (32 registers) in flight at a time, not all of them
    const float* w3l = w3tab + 256 * a + 32 * wave + 4 * hf;
    float q = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(w3l + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) q = __builtin_fmaf(w[e], aa[4 * g + e], q);
    }
    pp[a] = q;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) pp[a] = add32(pp[a]);
  if (hf == 0) {
    if constexpr (NA == 1) hp[j * 8] = pp[0];
    else {
#pragma unroll
      for (int a4 = 0; a4 < NA / 4; ++a4) {
        f32x4 o; o[0] = pp[4 * a4]; o[1] = pp[4 * a4 + 1]; o[2] = pp[4 * a4 + 2]; o[3] = pp[4 * a4 + 3];
        *reinterpret_cast<f32x4*>(hp + j * 8 + 4 * a4) = o;
      }
    }
  }
}
// h2 of the wave's 32 rows for the tile (in place of the summed accumulators) and its head partials into hp [sample 32][8]. NA = head rows computed
// (the W3ᵀ table is zero beyond n_act): a compile-time count keeps the sixteen table reads and the sums in one basic block
template <bool RAT, int NA>
__device__ __forceinline__ void rs_epilogue(f32x16& aa, float cs, const float* b2tab, const float* w3tab,
                                            float* hp, float* H2row, int wave, int j, int hf, bool st_ = false) {
  const float* b2l = b2tab + 32 * wave + 4 * hf;
  f32x4 bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(b2l + 8 * g);
#pragma unroll
  for (int i = 0; i < 16; ++i) aa[i] = __builtin_fmaf(aa[i], cs, bv[i >> 2][i & 3]);
  if (RAT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) aa[i] = tanh_fast(aa[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) aa[i] = __builtin_amdgcn_exp2f(aa[i] * TWO_LOG2E);
#pragma unroll
    for (int i = 0; i < 16; ++i) aa[i] = __builtin_amdgcn_rcpf(aa[i] + 1.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) aa[i] = __builtin_fmaf(-2.0f, aa[i], 1.0f);                // tanh_exp2, stage by stage
  }
  if (st_) RS_STAMP(6);
  if (H2row) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o; o[0] = aa[4 * g]; o[1] = aa[4 * g + 1]; o[2] = aa[4 * g + 2]; o[3] = aa[4 * g + 3];
      *reinterpret_cast<f32x4*>(H2row + 32 * wave + 4 * hf + 8 * g) = o;
    }
  }
  if (st_) RS_STAMP(7);
  rs_heads<NA>(aa, w3tab, hp, wave, j, hf);
}

// The forward pass of one network, persistent over 32-sample tiles, all eight waves in lock step. Stage s (one barrier each):
//   the matrix pipe   M(s): this wave's 32 rows x tile s, 48 MFMAs (+ 3 for layer 1 of tile s + 1 at the top)
//   the vector pipe   P(s + 1): the wave's h1 share of the next tile; E(s - 1): activation, h2 out, head partials of the previous tile; fold(s - 2)
// interleaved in ONE instruction stream — four groups of (four k-steps, four elements of P, four elements of E) — so that each wave's vector work runs in
// the shadow of its own and its SIMD partner's MFMAs. (The first form of this kernel gave the two waves of a SIMD opposite phases — one multiplies, one
// does all the vector work — with a barrier per phase: 0.82 ms per launch against the producer / consumer kernel's 0.58, the vector phase 3 µs per tile
// where 1.1 were issued: exposed LDS latency beside the partner's b128 stream, the layer-1 MFMAs queued behind the partner's 48, and every global load
// consumed behind the h2 stores' acknowledgements — vmcnt is in order. profiles/r06_c3_rs_forward.txt.)
// Observations never touch registers on their way in: 16-tile chunks are gathered by LDS-DMA (global_load_lds_dwordx4, lane = row piece) two chunks ahead,
// the row indices a chunk before that; the one s_waitcnt vmcnt(0) this needs sits at a chunk boundary — once per 16 tiles — not in every tile.
constexpr int R2_OFF_B1 = 2 * RS_XBYTES;                       // b1·2·log2(e) [256] f32
constexpr int R2_OFF_B2 = R2_OFF_B1 + 1024;                    // b2·2·log2(e) [256] f32
constexpr int R2_OFF_W3 = R2_OFF_B2 + 1024;                    // W3ᵀ [8][256] f32 (zero beyond n_act)
constexpr int R2_OFF_HP = R2_OFF_W3 + 8 * 1024;                // head partials [tile parity 2][wave 8][sample 32][8] f32
constexpr int R2_OFF_W1 = R2_OFF_HP + 2 * 8 * RS_MB * 8 * 4;   // W1 fragments (pack w1f) 16 KB
constexpr int R2_OFF_OBS = R2_OFF_W1 + 16384;                  // three observation chunks, [tile][piece][row 32][16 B]
constexpr int R2_OBS_BYTES = 16 * 32 * 8 * 4;                  // one chunk: 16 tiles of 8 floats per row, or 8 tiles of 16
constexpr int R2_LDS = R2_OFF_OBS + 3 * R2_OBS_BYTES;          // 159,744 bytes

template <int DP, int NA, bool STORE>
__device__ __forceinline__ void wide_rs_fwd_body(const FusedFwdPCArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  if (a.M <= 0) return;                                                 // (a one-network launch: the other half of the grid has nothing to do)
  constexpr int CH = DP == 8 ? 16 : 8;                                  // tiles per observation chunk
  constexpr int TB = 32 * DP * 4;                                       // bytes of one tile's observations
  constexpr int NP = DP / 4;                                            // 16-byte pieces per row
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int j = lane & 31, hf = lane >> 5;
  float* b1tab = reinterpret_cast<float*>(smx + R2_OFF_B1);
  float* b2tab = reinterpret_cast<float*>(smx + R2_OFF_B2);
  float* w3tab = reinterpret_cast<float*>(smx + R2_OFF_W3);
  float* hpall = reinterpret_cast<float*>(smx + R2_OFF_HP);
  if (tid < 256) { b1tab[tid] = a.W1f[4096 + tid]; b2tab[tid] = a.b2[tid] * TWO_LOG2E; }
  for (int i = tid; i < NA * 256; i += 512) w3tab[i] = i < a.A * 256 ? a.W3t[i] : 0.0f;
  for (int i = tid; i < 1024; i += 512) reinterpret_cast<f32x4*>(smx + R2_OFF_W1)[i] = reinterpret_cast<const f32x4*>(a.W1f)[i];
  RsWeights W;
  rs_load_weights(a.Wx2, wave, lane, W);
  const float w1un = a.w1sc[1];
  const float cs2 = a.wsc[1] * (1.0f / X2_ACT_SCALE) * TWO_LOG2E;       // accumulator -> 2·log2(e)·(W2·h1)
  const int ntiles = a.M / RS_MB;
  const int nblk = a.nblk > 0 ? a.nblk : (int)gridDim.x;
  const int nloc = ((int)blockIdx.x < ntiles && (int)blockIdx.x < nblk) ? (ntiles - (int)blockIdx.x + nblk - 1) / nblk : 0;
  if (nloc == 0) return;
  auto tile_of = [&](int i) { return (int)blockIdx.x + i * nblk; };
  // ---- observation chunks: wave w gathers tiles CH·c + w (+ 8): lane = 32·piece + row
  const int nper = CH / 8;                                              // tiles per wave and chunk
  auto row_index = [&](int i) { int g = tile_of(i) * RS_MB + j; g = g < a.M ? g : a.M - 1; return a.perm ? a.perm[g] : g; };
  int srcn[CH / 8];
  auto load_idx = [&](int c) {
#pragma unroll
    for (int q = 0; q < CH / 8; ++q) srcn[q] = row_index(CH * c + 8 * q + wave);
  };
  auto dma_chunk = [&](int c) {
    const unsigned base = lds_addr_of(smx + R2_OFF_OBS + (c % 3) * R2_OBS_BYTES);
#pragma unroll
    for (int q = 0; q < CH / 8; ++q) {
      const char* row = reinterpret_cast<const char*>(a.obs + (size_t)srcn[q] * (size_t)a.D);
#pragma unroll
      for (int pp = 0; pp < NP; pp += 2) {
        const int piece = pp + hf;
        if (4 * piece < a.D) lds_dma16_v(row + 16 * piece, base + (8 * q + wave) * TB + pp * 512);
      }
    }
  };
  (void)nper;
  load_idx(0);
  dma_chunk(0);
  load_idx(1);
  dma_chunk(1);
  load_idx(2);
  const float b3q = (lane & 7) < a.A ? a.b3[lane & 7] : 0.0f;           // the fold's bias, once (a load behind the h2 stores waits for their acknowledgements)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- helpers
  auto xfrag_of = [&](int i, P2& xb, float& xinv) {                     // observation fragment of local tile i from its chunk
    const unsigned char* tb = smx + R2_OFF_OBS + ((i / CH) % 3) * R2_OBS_BYTES + (i % CH) * TB;
    const int p0 = (8 * hf < DP) ? 2 * hf : 0;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(tb + p0 * 512 + 16 * j), q1 = *reinterpret_cast<const f32x4*>(tb + (p0 + 1) * 512 + 16 * j);
    float xr[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    rs_xfrag<DP>(xr, a.D, hf, xb, xinv, w1un);
  };
  auto fold = [&](int i) {                                              // wave w: samples 4w … 4w + 3 of local tile i, lanes 0-31 = (sample, head slot)
    if (lane < 32) {
      const int m = 4 * wave + (lane >> 3), q = lane & 7;
      if (q < a.A) {
        const float* hp = hpall + (i & 1) * 8 * (RS_MB * 8) + m * 8 + q;
        float z = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) z += hp[w8 * (RS_MB * 8)];
        a.Z[(size_t)a.ldz * (tile_of(i) * RS_MB + m) + q] = z + b3q;
      }
    }
  };
  // ---- prologue: h1 of tile 0, the observation fragment of tile 1
  f32x16 c1; float xinv;
  P2 xbn; float xinvn = 0.0f;
  {
    xfrag_of(0, xbn, xinv);
    c1 = rs_layer1(reinterpret_cast<const f16x8*>(smx + R2_OFF_W1) + (wave * 2) * 64 + 32 * hf + j, xbn);
    rs_produce<false>(c1, xinv, b1tab, reinterpret_cast<_Float16*>(smx), wave, j, hf);
    xfrag_of(1, xbn, xinvn);
  }
  f32x16 acc0, acc1;                                                    // M(s) accumulates into one while E(s - 1) reads the other: the roles swap every stage
#pragma unroll
  for (int q = 0; q < 16; ++q) acc1[q] = 0.0f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  auto stage = [&](int s, f32x16& acc, f32x16& ep) __attribute__((always_inline)) {
    asm volatile("" : "+v"(j), "+v"(hf));                               // per-lane addresses are formed per stage, not kept across the loop
    const bool st_ = s == 20;
    if (st_) RS_STAMP(0);
    if (s % CH == 0) {                                                  // chunk boundary: chunk s / CH + 1 has landed, the indices of + 2 are here
      const int c = s / CH;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      dma_chunk(c + 2);
      load_idx(c + 3);
    }
    if (s >= 2) fold(s - 2);                                            // (at the stage's tail instead: 526 vs 489 µs)
    // layer 1 of tile s + 1 (a tile beyond the block's last one is made from clamped rows and never read)
    xinv = xinvn;                                                       // (the fragment made here, at the stage's top, instead of a stage ahead: 489 vs 473 µs)
    c1 = rs_layer1(reinterpret_cast<const f16x8*>(smx + R2_OFF_W1) + (wave * 2) * 64 + 32 * hf + j, xbn);
    if (st_) RS_STAMP(1);
    const _Float16* Xs = reinterpret_cast<const _Float16*>(smx + (s & 1) * RS_XBYTES) + j * RS_ROW + 8 * hf;
    _Float16* Xn = reinterpret_cast<_Float16*>(smx + ((s + 1) & 1) * RS_XBYTES) + j * RS_ROW + 32 * wave + 4 * hf;
    const float* b1l = b1tab + 32 * wave + 4 * hf;
    const float* b2l = b2tab + 32 * wave + 4 * hf;
    const float* w3l = w3tab + 32 * wave + 4 * hf;
    // E(s - 1) goes to the previous tile's rows; in stage 0 there is none: the stores land on tile 0's own rows and stage 1 overwrites them
    const int te = s >= 1 ? s - 1 : 0;
    float* H2row = STORE ? a.H2 + (size_t)256 * (tile_of(te) * RS_MB + j) + 32 * wave + 4 * hf : nullptr;
    float pp[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) pp[q] = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    // One group = four k-steps of M(s) (twelve MFMAs, ONE dependent chain: a single accumulator) + four elements each of P(s + 1) and E(s - 1) (~150 vector
    // instructions). Issued in the compiler's order — the chain first — the chain stalls the wave 32+ cycles per MFMA while the vector pipe idles, then the
    // vector block runs with the matrix pipe idle, and with both waves of a SIMD in lock step neither hides the other's: 3.8 µs per stage against 2.0 of vector
    // and 1.5 of matrix work. So the SIMD partners run the two halves in OPPOSITE order: waves 0-3 chain then vector block, waves 4-7 vector block then chain.
    auto mpart = [&](int g) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int ks = 4 * g + kk;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(Xs + 16 * ks), bl = *reinterpret_cast<const f16x8*>(Xs + RS_MB * RS_ROW + 16 * ks);
        acc = mfma_f16(W.wl[ks], bh, acc);                                // (a second accumulator for the hi·hi terms — independent neighbours — measured equal: 483 vs 482 µs)
        acc = mfma_f16(W.wh[ks], bl, acc);
        acc = mfma_f16(W.wh[ks], bh, acc);
      }
    };
    auto vpart = [&](int g) __attribute__((always_inline)) {
      const f32x4 bv1 = *reinterpret_cast<const f32x4*>(b1l + 8 * g), bv2 = *reinterpret_cast<const f32x4*>(b2l + 8 * g);
      // P(s + 1), elements 4g … 4g + 3
      {
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(c1[4 * g + e], xinv, bv1[e]));
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_amdgcn_rcpf(hv[e] + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_fmaf(-2.0f * X2_ACT_SCALE, hv[e], X2_ACT_SCALE);
        uint2 hh, ll;
        split2x4(hv, 1.0f, hh, ll);
        *reinterpret_cast<uint2*>(Xn + 8 * g) = hh;
        *reinterpret_cast<uint2*>(Xn + RS_MB * RS_ROW + 8 * g) = ll;
      }
      // E(s - 1), elements 4g … 4g + 3
      {
        f32x4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(ep[4 * g + e], cs2, bv2[e]));
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_amdgcn_rcpf(hv[e] + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[e] = __builtin_fmaf(-2.0f, hv[e], 1.0f);
        if (STORE) *reinterpret_cast<f32x4*>(H2row + 8 * g) = hv;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(w3l + 256 * q + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) pp[q] = __builtin_fmaf(w[e], hv[e], pp[q]);
        }
      }
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      mpart(g);
      vpart(g);
      if (g == 2) xfrag_of(s + 2, xbn, xinvn);                          // the tile after next: its fragment is ready when the next stage opens
      __builtin_amdgcn_sched_barrier(0);
    }
    if (st_) RS_STAMP(2);
#pragma unroll
    for (int q = 0; q < NA; ++q) pp[q] = add32(pp[q]);
    if (hf == 0) {
      float* hp = hpall + ((te & 1) * 8 + wave) * (RS_MB * 8) + j * 8;
      if constexpr (NA == 1) hp[0] = pp[0];
      else {
#pragma unroll
        for (int a4 = 0; a4 < NA / 4; ++a4) {
          f32x4 o; o[0] = pp[4 * a4]; o[1] = pp[4 * a4 + 1]; o[2] = pp[4 * a4 + 2]; o[3] = pp[4 * a4 + 3];
          *reinterpret_cast<f32x4*>(hp + 4 * a4) = o;
        }
      }
    }
    if (st_) RS_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st_) RS_STAMP(4);
  };
#pragma unroll 1
  for (int s = 0; s <= nloc; ++s) {
    stage(s, acc0, acc1);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc0[q];                     // (two copies of the stage with the accumulators' roles swapped instead: the same 492 µs)
  }
  if (nloc >= 1) fold(nloc - 1);
  // (stage 0's head partials of "tile -1" went to parity 0 of tile 0's slot … and were rewritten by stage 1 before any fold read them)
}

// NA: head rows the actor's blocks compute (4 or 8); the critic's compute one
template <int DP, int NA, bool STORE>
__global__ void __launch_bounds__(512) wide_rs_fwd_kernel(FusedFwdPCArgs a0, FusedFwdPCArgs a1) {
  if (blockIdx.y == 0) wide_rs_fwd_body<DP, NA, STORE>(a0); else wide_rs_fwd_body<DP, 1, STORE>(a1);
}

// ======================================================================================================================================
// The backward pass of one network with W2ᵀ register-stationary (option wide_rs bit 3; replaces wide_fused_bwd_kernel in its split-plane flavour).
// The same lock-step stage as the forward: wave w holds the fp16x2 fragments of W2ᵀ for hidden units v = 32w … 32w + 31 (pack x2b, 128 registers) and uses
// them as the B operand — mfma(δ2 fragment, W2ᵀ fragment) — so that the product comes out TRANSPOSED: rows (registers) = samples, columns (lanes) = units,
// and the sums over samples that dW1 / db1 need are per-lane sums (8 + 1 accumulators per lane, no cross-lane work until the launch ends). Per 32-sample stage s:
//   M(s)      δ1ᵀ-pre = δ2(s)ᵀ·W2ᵀ: 48 MFMAs, A-fragments from the δ2 tile in LDS;
//   P(s + 1)  the wave's 32 units of the next tile's δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²) — h2 arrives by LDS-DMA two tiles ahead, δ3 (A floats per sample) in registers — scaled
//             per sample into the fp16 window (bound Σ|δ3|·max|W3|, as before), split, into the δ2 tile AND out to the f16 planes the weight-gradient kernel reads;
//   E(s - 1)  h1 of the previous tile recomputed on the matrix pipe (one fp16x2 product per wave, operands swapped like the main product), 1 − h1² = 4r(1 − r),
//             δ1 = acc·(4/(s1·scale))·r(1 − r), db1 += δ1, dW1[c] += δ1·x[c] with the sample's observation broadcast from LDS.
// h2 tiles: [sample][256] f32 rows of 1 KB at a pitch of 1040 B (conflict-free 16-byte reads), two buffers; one `s_waitcnt vmcnt(4)` per stage waits for the
// tile's four LDS-DMA pieces per wave and leaves the stage's last four plane stores in flight.
// ======================================================================================================================================
#ifndef CRL_RSB_GW_MFMA
#define CRL_RSB_GW_MFMA 0   // dW1 / db1 of the backward stage: 1 = on the matrix pipe (v_mfma_f32_32x32x2_f32, the observation as the B operand) — correct, and SLOWER: 794 vs 718 µs
                           // per launch (16 Float32 MFMAs of 64 cycles per stage cost the matrix pipe more than 144 FMAs cost the vector pipe); 0 = 8 + 1 FMAs per element
#endif
#ifndef CRL_RSB_SPLIT
#define CRL_RSB_SPLIT 1   // scheduling barriers inside a group of the backward stage: bit 0 between the MFMA chain and P, bit 1 between P and E (none: ~50 B of scratch, 823 µs;
                          // both 716-720; bit 0 alone 711-713; bit 1 alone 728-740)
#endif
constexpr int RB_HROW = 1040;
constexpr int RB_HBYTES = RS_MB * RB_HROW;                     // 33,280: one h2 tile
constexpr int RB_OFF_H = 2 * RS_XBYTES;                        // two δ2 tiles first
constexpr int RB_OFF_W3 = RB_OFF_H + 2 * RB_HBYTES;            // W3ᵀ [8][256] f32 (zero beyond n_act)
constexpr int RB_OFF_XS = RB_OFF_W3 + 8192;                    // observations [tile parity 2][piece 4][sample 32][4] f32
constexpr int RB_OFF_DZ = RB_OFF_XS + 2 * 4 * RS_MB * 16;      // head cotangents [tile & 3][piece 2][sample 32][4] f32
constexpr int RB_OFF_TAB = RB_OFF_DZ + 4 * 2 * RS_MB * 16;     // per wave: xinv [32], then inv4 [tile & 3][32] f32
constexpr int RB_LDS = RB_OFF_TAB + 8 * 5 * RS_MB * 4;         // 155,648 bytes
static_assert(RB_LDS <= 160 * 1024, "register-stationary backward: LDS budget");

template <int DP, int NA, bool DW3>
__device__ __forceinline__ void wide_rs_bwd_body(const FusedBwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  if (a.M <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int j = lane & 31, hf = lane >> 5;
  float* w3tab = reinterpret_cast<float*>(smx + RB_OFF_W3);
  float* xsall = reinterpret_cast<float*>(smx + RB_OFF_XS);
  float* dzall = reinterpret_cast<float*>(smx + RB_OFF_DZ);
  float* tabw = reinterpret_cast<float*>(smx + RB_OFF_TAB) + wave * (5 * RS_MB);    // this wave's private tables
  for (int i = tid; i < (RB_LDS - RB_OFF_XS) / 4; i += 512) xsall[i] = 0.0f;           // observations, cotangents and tables: stage 0 "finishes" a tile that does not exist
  for (int i = tid; i < NA * 256; i += 512) w3tab[i] = i < a.A * 256 ? a.W3t[i] : 0.0f;
  RsWeights W;
  rs_load_weights(a.Wx2b, wave, lane, W);
  P2 w1f;
  { const f16x8* w1p = reinterpret_cast<const f16x8*>(a.W1f) + (wave * 2) * 64 + lane; w1f.hi = w1p[0]; w1f.lo = w1p[64]; }
  const float b1v = a.W1s[256 * DP + 32 * wave + j];                   // 2·log2(e)·b1 of this lane's unit
  const float w1un = a.w1sc[1];
  const float wun4 = 4.0f * a.wsc[1];
  float wmaxr[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) wmaxr[q] = q < a.A ? a.wmax[q] : 0.0f;
  const int ntiles = a.M / RS_MB;
  const int nblk = a.nblk > 0 ? a.nblk : (int)gridDim.x;
  const int nloc = ((int)blockIdx.x < ntiles && (int)blockIdx.x < nblk) ? (ntiles - (int)blockIdx.x + nblk - 1) / nblk : 0;
  if (nloc == 0) return;
  auto tile_of = [&](int i) { return (int)blockIdx.x + (i < nloc ? i : nloc - 1) * nblk; };   // tiles beyond the block's last are made from its last one and never used
  float gW1[DP], gB1 = 0.0f, gW3[NA];
#pragma unroll
  for (int c = 0; c < DP; ++c) gW1[c] = 0.0f;
  // dW1 / db1 on the matrix pipe (CRL_RSB_GW_MFMA): C[unit][column] += δ1[unit][sample]·B[sample][column] with B = (observation | 1 | 0 …), K = the two samples a register
  // holds across the lane halves — v_mfma_f32_32x32x2_f32 takes δ1 as it sits in the registers (Float32, no split). Lane = column here: columns 0 … obs_dim - 1 are dW1, column
  // DP is db1, the others stay zero; 16 registers replace the 8 + 1 per-lane sums, 16 MFMAs and 16 four-byte LDS reads per stage replace 144 FMAs and 32 broadcast reads
  f32x16 gWm;
#pragma unroll
  for (int q = 0; q < 16; ++q) gWm[q] = 0.0f;
#pragma unroll
  for (int q = 0; q < NA; ++q) gW3[q] = 0.0f;
  // dW3[a][v] += δ3[a][sample]·h2[v][sample] for four samples (rows 8g + 4hf + e) of tile i: lane = unit v like the epilogue — the tile's h2 and δ3 are
  // both in LDS while its δ2 is staged, so the two dW3 sweeps over h2 (1.07 GB per optimiser step, 2 x 0.11 ms alone) need not be launched. Template flag DW3
  // (option wide_rs bit 4, on by default). Placement matters: inside the groups its reads and sums spilled ~20 registers (865 vs 665 µs for this kernel — more
  // than the 117 µs the co-running sweeps cost the weight-gradient kernel); as four scheduling regions BEHIND the groups: 740 vs 691 µs, weight gradient
  // 580 -> 465 µs, iteration 34.6 -> 32.5 ms on one box
  auto dw3_part = [&](int i, int g) __attribute__((always_inline)) {
    const float* hcol = reinterpret_cast<const float*>(smx + RB_OFF_H + (i & 1) * RB_HBYTES) + 32 * wave + j;
    const float* dzt = dzall + (i & 3) * (2 * RS_MB * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int sidx = 8 * g + 4 * hf + e;
      const float hv = hcol[sidx * (RB_HROW / 4)];
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(dzt + sidx * 4);
      if constexpr (NA == 1) gW3[0] = __builtin_fmaf(z0[0], hv, gW3[0]);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) gW3[q] = __builtin_fmaf(z0[q], hv, gW3[q]);
        if constexpr (NA == 8) {
          const f32x4 z1 = *reinterpret_cast<const f32x4*>(dzt + RS_MB * 4 + sidx * 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) gW3[4 + q] = __builtin_fmaf(z1[q], hv, gW3[4 + q]);
        }
      }
    }
  };
  // ---- input pipelines
  auto dma_h2 = [&](int i) {                                            // tile i's h2 rows 4·wave … 4·wave + 3 into buffer i & 1
    const float* src = a.H2 + (size_t)256 * (tile_of(i) * RS_MB + 4 * wave);
    const unsigned base = lds_addr_of(smx + RB_OFF_H + (i & 1) * RB_HBYTES) + 4 * wave * RB_HROW;
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_dma16(src + 256 * r, lane * 16, base + r * RB_HROW);
  };
  // δ3 and the observations come by per-lane LDS-DMA as well (lane = 32·piece + sample): no registers hold inputs across a stage
  auto dma_dz = [&](int i) {                                            // wave 1: the tile's head cotangents, pieces of 4 floats
    if (4 * hf < NA) {
      const float* src = a.dZ + (size_t)a.ldd * (tile_of(i) * RS_MB + j) + 4 * hf;
      lds_dma16_v(src, lds_addr_of(dzall + (i & 3) * (2 * RS_MB * 4)));
    }
  };
  int srcn = 0;
  auto row_index = [&](int i) { int g = tile_of(i) * RS_MB + j; return a.perm ? a.perm[g] : g; };
  auto dma_obs = [&](int i) {                                           // wave 0: the tile's observation rows through the permutation (index in srcn), 4-float pieces
    const char* row = reinterpret_cast<const char*>(a.obs + (size_t)srcn * (size_t)a.D);
#pragma unroll
    for (int pp = 0; pp < DP / 4; pp += 2) {
      const int piece = pp + hf;
      if (4 * piece < a.D) lds_dma16_v(row + 16 * piece, lds_addr_of(xsall + (i & 1) * (4 * RS_MB * 4)) + pp * 512);
    }
  };
  auto load_dz = [&](int i, float (&o)[NA]) {                           // this lane's sample's cotangents from the LDS copy
    const float* p = dzall + (i & 3) * (2 * RS_MB * 4) + j * 4;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
    if constexpr (NA == 1) o[0] = v0[0];
    else {
      o[0] = v0[0]; o[1] = v0[1]; o[2] = v0[2]; o[3] = v0[3];
      if constexpr (NA == 8) { const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + RS_MB * 4); o[4] = v1[0]; o[5] = v1[1]; o[6] = v1[2]; o[7] = v1[3]; }
#pragma unroll
      for (int q = 0; q < NA; ++q) o[q] = q < a.A ? o[q] : 0.0f;
    }
  };
  // ---- P: this wave's 32 units (lane: sample j, units 32·wave + 16·hf + 0 … 15) of tile i's δ2
  auto stage_d2 = [&](int i, bool store) {
    float dzv[NA];
    load_dz(i, dzv);
    float bound = 0.0f;
#pragma unroll
    for (int q = 0; q < NA; ++q) bound = __builtin_fmaf(__builtin_fabsf(dzv[q]), wmaxr[q], bound);
    float s1, i1;
    pow2_scale(bound, s1, i1);
    if (hf == 0) tabw[RS_MB + (i & 3) * RS_MB + j] = i1 * wun4;
    const int gm = tile_of(i) * RS_MB + j;
    if (store && wave == 0 && hf == 0) a.d2s[gm] = i1;
    const unsigned char* hb = smx + RB_OFF_H + (i & 1) * RB_HBYTES + j * RB_HROW + (32 * wave + 16 * hf) * 4;
    _Float16* xo = reinterpret_cast<_Float16*>(smx + (i & 1) * RS_XBYTES) + j * RS_ROW + 32 * wave + 16 * hf;
    _Float16* dh = a.D2h + (size_t)256 * gm + 32 * wave + 16 * hf;
    uint2 hp, lp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 hq = *reinterpret_cast<const f32x4*>(hb + 16 * g);
      f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3tab + 256 * q + 32 * wave + 16 * hf + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(w[e], dzv[q], d[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = d[e] * __builtin_fmaf(-hq[e], hq[e], 1.0f);
      uint2 hh, ll;
      split2x4(d, s1, hh, ll);
      *reinterpret_cast<uint2*>(xo + 4 * g) = hh;
      *reinterpret_cast<uint2*>(xo + RS_MB * RS_ROW + 4 * g) = ll;
      if (g & 1) {
        if (store) {
          uint4 oh, ol; oh.x = hp.x; oh.y = hp.y; oh.z = hh.x; oh.w = hh.y; ol.x = lp.x; ol.y = lp.y; ol.z = ll.x; ol.w = ll.y;
          *reinterpret_cast<uint4*>(dh + 4 * (g - 1)) = oh;
          *reinterpret_cast<uint4*>(dh + (size_t)256 * a.M + 4 * (g - 1)) = ol;
        }
      } else { hp = hh; lp = ll; }
    }
  };
  // ---- prologue: h2 and δ3 of tiles 0 and 1, the row indices of tile 0
  dma_h2(0); dma_h2(1);
  if (wave == 1) { dma_dz(0); dma_dz(1); }
  if (wave == 0) srcn = row_index(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stage_d2(0, true);
#pragma unroll
  for (int g = 0; g < 4; ++g) if constexpr (DW3) dw3_part(0, g);
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc1[q] = 0.0f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // pst: tile s + 1 exists (its δ2 planes and scale are stored). A compile-time flag: a branch around the stores inside the groups ends the basic block,
  // and the compiler then gathers ALL of E behind the last MFMA (nothing overlaps, 120 registers spill)
  auto stage = [&](int s, f32x16& acc, const f32x16& ep, auto pst_) __attribute__((always_inline)) {
    constexpr bool pst = decltype(pst_)::value;
    asm volatile("" : "+v"(j), "+v"(hf));
    // inputs, all by LDS-DMA: h2 of tile s + 2 (buffer s & 1: its last reader was P(s) a stage ago), δ3 of tile s + 2, the observations of tile s
    // (read from the next stage on: E(s) and its layer 1), the row indices of tile s + 1
    dma_h2(s + 2);
    if (wave == 1) dma_dz(s + 2);
    if (wave == 0) { dma_obs(s); srcn = row_index(s + 1); }
    float dz[NA];
    load_dz(s + 1, dz);
    // layer 1 of tile s - 1 on the matrix pipe, operands swapped: rows = samples
    f32x16 c1;
    {
      const float* xr = xsall + ((s + 1) & 1) * (4 * RS_MB * 4) + ((8 * hf < DP) ? 2 * hf : 0) * (RS_MB * 4) + j * 4;
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(xr), q1 = *reinterpret_cast<const f32x4*>(xr + RS_MB * 4);
      float xv[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      P2 xb; float xinv;
      rs_xfrag<DP>(xv, a.D, hf, xb, xinv, w1un);
      if (hf == 0) tabw[j] = xinv;
#pragma unroll
      for (int r = 0; r < 16; ++r) c1[r] = 0.0f;
      c1 = mfma_x2(xb, w1f, c1);
    }
    const _Float16* Xs = reinterpret_cast<const _Float16*>(smx + (s & 1) * RS_XBYTES) + j * RS_ROW + 8 * hf;
    const float* xe = xsall + ((s + 1) & 1) * (4 * RS_MB * 4);         // observations of tile s - 1, [piece][sample][4]
    const float* inv4t = tabw + RS_MB + ((s + 3) & 3) * RS_MB;         // (s - 1) & 3
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    // P(s + 1) runs in four parts beside the groups below: its state
    float bound = 0.0f;
#pragma unroll
    for (int q = 0; q < NA; ++q) bound = __builtin_fmaf(__builtin_fabsf(dz[q]), wmaxr[q], bound);
    float s1, i1;
    pow2_scale(bound, s1, i1);
    if (hf == 0) tabw[RS_MB + ((s + 1) & 3) * RS_MB + j] = i1 * wun4;
    const int gmn = tile_of(s + 1) * RS_MB + j;
    if (pst && wave == 0 && hf == 0) a.d2s[gmn] = i1;
    const unsigned char* hb = smx + RB_OFF_H + ((s + 1) & 1) * RB_HBYTES + j * RB_HROW + (32 * wave + 16 * hf) * 4;
    _Float16* xo = reinterpret_cast<_Float16*>(smx + ((s + 1) & 1) * RS_XBYTES) + j * RS_ROW + 32 * wave + 16 * hf;
    _Float16* dh = a.D2h + (size_t)256 * gmn + 32 * wave + 16 * hf;
    uint2 hp, lp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int ks = 4 * g + kk;
        P2 xa;
        xa.hi = *reinterpret_cast<const f16x8*>(Xs + 16 * ks); xa.lo = *reinterpret_cast<const f16x8*>(Xs + RS_MB * RS_ROW + 16 * ks);
        acc = mfma_f16(xa.lo, W.wh[ks], acc);
        acc = mfma_f16(xa.hi, W.wl[ks], acc);
        acc = mfma_f16(xa.hi, W.wh[ks], acc);
      }
      if (CRL_RSB_SPLIT & 1) __builtin_amdgcn_sched_barrier(0);
      // P(s + 1), units 4g … 4g + 3 of this lane's sixteen
      {
        const f32x4 hq = *reinterpret_cast<const f32x4*>(hb + 16 * g);
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < NA; ++q) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(w3tab + 256 * q + 32 * wave + 16 * hf + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(w[e], dz[q], d[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = d[e] * __builtin_fmaf(-hq[e], hq[e], 1.0f);
        uint2 hh, ll;
        split2x4(d, s1, hh, ll);
        *reinterpret_cast<uint2*>(xo + 4 * g) = hh;
        *reinterpret_cast<uint2*>(xo + RS_MB * RS_ROW + 4 * g) = ll;
        if (g & 1) {
          if (pst) {
            uint4 oh, ol; oh.x = hp.x; oh.y = hp.y; oh.z = hh.x; oh.w = hh.y; ol.x = lp.x; ol.y = lp.y; ol.z = ll.x; ol.w = ll.y;
            *reinterpret_cast<uint4*>(dh + 4 * (g - 1)) = oh;
            *reinterpret_cast<uint4*>(dh + (size_t)256 * a.M + 4 * (g - 1)) = ol;
          }
        } else { hp = hh; lp = ll; }
      }
      if (CRL_RSB_SPLIT & 2) __builtin_amdgcn_sched_barrier(0);
      // E(s - 1), samples 8g + 4hf + 0 … 3 of this lane's unit
      {
        const f32x4 xiq = *reinterpret_cast<const f32x4*>(tabw + 8 * g + 4 * hf), ivq = *reinterpret_cast<const f32x4*>(inv4t + 8 * g + 4 * hf);
        f32x4 d1;
#pragma unroll
        for (int e = 0; e < 4; ++e) d1[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(c1[4 * g + e], xiq[e], b1v));
#pragma unroll
        for (int e = 0; e < 4; ++e) d1[e] = __builtin_amdgcn_rcpf(d1[e] + 1.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) d1[e] = (ep[4 * g + e] * ivq[e]) * __builtin_fmaf(-d1[e], d1[e], d1[e]);   // acc·(4/(s1·scale))·r(1 − r)
        if constexpr (CRL_RSB_GW_MFMA != 0) {
          const int cc = j;                                              // this lane's column of the dW1 | db1 tile
          const float* xcol = xe + (cc >> 2) * (RS_MB * 4) + (cc & 3) + (8 * g + 4 * hf) * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bx = cc < DP ? xcol[4 * e] : (cc == DP ? 1.0f : 0.0f);
            gWm = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[e], bx, gWm, 0, 0, 0);
          }
        } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gB1 += d1[e];
          const float* xs = xe + (8 * g + 4 * hf + e) * 4;
#pragma unroll
          for (int c4 = 0; c4 < DP / 4; ++c4) {
            const f32x4 xq = *reinterpret_cast<const f32x4*>(xs + c4 * (RS_MB * 4));
#pragma unroll
            for (int c = 0; c < 4; ++c) gW1[4 * c4 + c] = __builtin_fmaf(d1[e], xq[c], gW1[4 * c4 + c]);
          }
        }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (pst && DW3) {                                         // behind the groups, in regions of its own (a ROLLED loop here: 580 B of scratch)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) { dw3_part(s + 1, g); __builtin_amdgcn_sched_barrier(0); }
    }
    // the h2 pieces of tile s + 2 have landed (only this stage's last four plane stores may still be in flight), LDS writes are done
    if (pst) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");      // (no plane stores behind the pieces in the block's last stages)
    __builtin_amdgcn_s_barrier();
  };
#pragma unroll 1
  for (int s = 0; s + 1 < nloc; ++s) {
    stage(s, acc0, acc1, std::true_type{});
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc0[q];
  }
  for (int s = nloc - 1; s <= nloc; ++s) {                              // the block's last tile is multiplied, then finished
    stage(s, acc0, acc1, std::false_type{});
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc0[q];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // no LDS-DMA piece may land after the block has given its LDS back
  // ---- the block's dW1 / db1 partial: the two lane halves hold different samples of the same unit
  if constexpr (CRL_RSB_GW_MFMA != 0) {                                  // lane = column, registers = this half's 16 units: 8·(r >> 2) + 4·hf + (r & 3)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int v = 32 * wave + 8 * (r >> 2) + 4 * hf + (r & 3);
      if (j < a.D) a.pW1[(size_t)blockIdx.x * 256 * a.D + v + 256 * j] = gWm[r];
      else if (j == DP) a.pB1[(size_t)blockIdx.x * 256 + v] = gWm[r];
    }
  }
  gB1 = add32(gB1);
#pragma unroll
  for (int c = 0; c < DP; ++c) gW1[c] = add32(gW1[c]);
#pragma unroll
  for (int q = 0; q < NA; ++q) gW3[q] = add32(gW3[q]);
  if (hf == 0) {
    const int v = 32 * wave + j;
#pragma unroll
    for (int q = 0; q < NA; ++q) if (DW3 && q < a.A) a.pW3[(size_t)blockIdx.x * 256 * a.A + (size_t)v * a.A + q] = gW3[q];
    if constexpr (CRL_RSB_GW_MFMA == 0) {
      a.pB1[(size_t)blockIdx.x * 256 + v] = gB1;
#pragma unroll
      for (int c = 0; c < DP; ++c) if (c < a.D) a.pW1[(size_t)blockIdx.x * 256 * a.D + v + 256 * c] = gW1[c];
    }
  }
}

template <int DP, int NA0, bool DW3>
__global__ void __launch_bounds__(512) wide_rs_bwd_kernel(FusedBwdArgs a0, FusedBwdArgs a1) {
  if (blockIdx.y == 0) wide_rs_bwd_body<DP, NA0, DW3>(a0); else wide_rs_bwd_body<DP, 1, DW3>(a1);
}

// ======================================================================================================================================
// The rollout with the ACTOR register-stationary (ppo.jl:123-166; option wide_rs bit 1). wide_rollout_pc_kernel walks 16 weight slabs per step —
// both networks, 32 KB each through LDS-DMA with a barrier per slab — and a step costs 30 µs at C3, almost all of it the latency of the slab
// stream. The env loop only needs the ACTOR: the critic's values are stored for the GAE pass and nothing in the loop reads them, so the critic
// leaves the loop (one batched forward over the stored observations behind the rollout: wide_rollout in wide.hip) and the actor's 256 KB of
// fp16x2 W2 pieces live in the eight waves' registers for all num_steps steps. A block owns 64 envs; per step
//   P  every wave: layer 1 of its 32 units for the 64 envs (one fp16x2 product per 32-env half, tanh_fast, split, into the h1 tile in LDS)
//   M  every wave: its 32 rows x 64 envs, 96 MFMAs, B-fragments from the h1 tile
//   E  tanh_fast, head partials of its rows into LDS
//   S  wave 0, lane = env: the eight partials + b3 in fixed order, softmax / sample / env step / Buffer.add! (wide_step_env), the next observation
//      into LDS for everybody
// with three barriers. The actor keeps the reference's rational tanh_fast in both layers, as in the other rollout kernels: its logits decide
// action indices.
// ======================================================================================================================================
constexpr int RR_MB = 64;
constexpr int RR_XBYTES = 2 * RR_MB * RS_ROW * 2;              // 67,584: the h1 tile, [piece][env][RS_ROW halves]
constexpr int RR_OFF_B1 = RR_XBYTES;
constexpr int RR_OFF_W3 = RR_OFF_B1 + 1024;
constexpr int RR_OFF_B2 = RR_OFF_W3 + 8 * 1024;
constexpr int RR_OFF_HP = RR_OFF_B2 + 1024;                    // head partials [wave 8][env 64][8] f32
constexpr int RR_OFF_W1 = RR_OFF_HP + 8 * RR_MB * 8 * 4;       // W1 fragments (pack w1f) 16 KB
constexpr int RR_OFF_OBS = RR_OFF_W1 + 16384;                  // current observations [env 64][16] f32
constexpr int RR_OFF_ST = RR_OFF_OBS + RR_MB * 16 * 4;         // per-env state [env 64][8] words: next_done, ep_return, ep_length, env_t, env_state[0..3]
constexpr int RR_OFF_B3 = RR_OFF_ST + RR_MB * 8 * 4;           // b3 [8] f32
constexpr int RR_LDS = RR_OFF_B3 + 32;                         // 116,768 bytes
struct RsRollArgs { RollPCNet n; WStepArgs s; int D; };

// softmax_rt / sample_rt / pick_rt of wide.hip on N <= 8 slots (same operation order)
template <int N>
__device__ __forceinline__ void softmax_n(const float (&z)[N], int A, float (&p)[N], float (&lp)[N]) {
  float m = z[0];
#pragma unroll
  for (int a = 1; a < N; ++a) if (a < A) m = fmaxf(m, z[a]);
  float sm = 0.0f;
#pragma unroll
  for (int a = 0; a < N; ++a) if (a < A) { p[a] = expf(z[a] - m); sm += p[a]; }
#pragma unroll
  for (int a = 0; a < N; ++a) if (a < A) p[a] = p[a] / sm;
  float ls = 0.0f;
#pragma unroll
  for (int a = 0; a < N; ++a) if (a < A) { lp[a] = z[a] - m; ls += expf(lp[a]); }
  const float l = logf(ls);
#pragma unroll
  for (int a = 0; a < N; ++a) if (a < A) lp[a] = lp[a] - l;
}
template <int N>
__device__ __forceinline__ int sample_n(const float (&p)[N], int A, double u) {
  float sw = 0.0f;
#pragma unroll
  for (int a = 0; a < N; ++a) if (a < A) sw += p[a];
  const double t = u * (double)sw;
  int i = 0;
  float cw = p[0];
#pragma unroll
  for (int a = 1; a < N; ++a) {
    const bool go = (a < A) && ((double)cw < t) && (i == a - 1);
    i = go ? a : i;
    cw = go ? cw + p[a] : cw;
  }
  return i;
}
template <int N>
__device__ __forceinline__ float pick_n(const float (&v)[N], int A, int i) {
  float r = v[0];
#pragma unroll
  for (int a = 1; a < N; ++a) if (a < A) r = (i == a) ? v[a] : r;
  return r;
}

// wide_step_env (ppo.jl:125-165) with the env's running state in LDS instead of global memory: a step then LOADS nothing from HBM / L2 — in
// wide_step_env the dozen dependent loads (episode counters, env state, the observation to copy) were 3-4 of a step's 5.4 µs behind the previous
// step's stores (vmcnt is in order). obs_l: the env's current observation [16] (in: what the buffer keeps, out: the next one); st_l: its 8 state words.
template <int NA>
__device__ __forceinline__ void rs_step_env(const WStepArgs& a, int e, int step, const float (&z)[NA], float* obs_l, float* st_l,
                                            double& st_n, double& st_ret, double& st_len, double& st_max) {
  const DevCfg& c = a.c;
  const int D = c.D, A = c.A;
  const uint32_t gid = c.env_id_offset + (uint32_t)e;
  const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
  const size_t b = (size_t)e + (size_t)c.nt * step;
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(st_l), s1 = *reinterpret_cast<const f32x4*>(st_l + 4);
  f32x4 x[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(obs_l + 4 * q);
  const int nd = __float_as_int(s0[0]);
  int ep_len = __float_as_int(s0[2]) + 1;                            // ppo.jl:125
  float p[NA], lp[NA];
  softmax_n<NA>(z, A, p, lp);                                        // ppo.jl:127 get_action
  const double u = u53(philox_env(c.seed, gid, gstep, 0));
  const int act = sample_n<NA>(p, A, u);
  const float lpa = pick_n<NA>(lp, A, act);
  float* ob = a.obs + b * (size_t)D;
  if ((D & 3) == 0) {                                                // ppo.jl:133-140 Buffer.add!
#pragma unroll
    for (int q = 0; q < 4; ++q) if (4 * q < D) *reinterpret_cast<f32x4*>(ob + 4 * q) = x[q];
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) if (4 * q + i < D) ob[4 * q + i] = x[q][i];
  }
  a.action[b] = act; a.logprob[b] = lpa; a.terminal[b] = (uint8_t)nd;
  bool done; float rew;
  f32x4 es = s1; int t_env = __float_as_int(s0[3]);
  if (c.env_kind == CRL_ENV_CARTPOLE) {
    float s[4] = {s1[0], s1[1], s1[2], s1[3]};
    done = cartpole_step(s, t_env, act);                             // ppo.jl:130
    rew = done ? 0.0f : 1.0f;                                        // ppo.jl:132
    float so[4] = {s[0], s[1], s[2], s[3]};                          // ppo.jl:143: the observation is taken before the reset (Q7)
    if (done) {
      cartpole_reset(s, c.seed, gid, gstep, 1);                      // ppo.jl:164
      t_env = 0;
      if (!c.stale_obs) for (int i = 0; i < 4; ++i) so[i] = s[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[0][i] = so[i]; es[i] = s[i]; }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * q < D) {
        float o4[4];
        synth_obs4(c.seed, gid, gstep, q, o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[q][i] = 4 * q + i < D ? o4[i] : 0.0f;
      }
    synth_reward_done(c.seed, gid, gstep, rew, done);
  }
  a.reward[b] = rew;
  float ep_ret = s0[1] + rew;                                        // ppo.jl:145
  if (done) {                                                        // ppo.jl:147-165
    st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, fmax(0.0, (double)ep_ret));
    if (a.ring_cap > 0) {
      const uint32_t slot = atomicAdd(a.ring_count, 1u);
      if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
    }
    ep_ret = 0.0f; ep_len = 0;
  }
  f32x4 n0; n0[0] = __int_as_float(done ? 1 : 0); n0[1] = ep_ret; n0[2] = __int_as_float(ep_len); n0[3] = __int_as_float(t_env);   // ppo.jl:144
  *reinterpret_cast<f32x4*>(st_l) = n0; *reinterpret_cast<f32x4*>(st_l + 4) = es;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(obs_l + 4 * q) = x[q];
}

template <int DP, int NA>
__global__ void __launch_bounds__(512) wide_rs_rollout_kernel(RsRollArgs r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int j = lane & 31, hf = lane >> 5;
  const int m0 = blockIdx.x * RR_MB;
  const RollPCNet& nn = r.n;
  float* b1tab = reinterpret_cast<float*>(smx + RR_OFF_B1);
  float* w3tab = reinterpret_cast<float*>(smx + RR_OFF_W3);
  float* b2tab = reinterpret_cast<float*>(smx + RR_OFF_B2);
  float* hpall = reinterpret_cast<float*>(smx + RR_OFF_HP);
  float* obsl = reinterpret_cast<float*>(smx + RR_OFF_OBS);
  float* stl = reinterpret_cast<float*>(smx + RR_OFF_ST);
  float* b3l = reinterpret_cast<float*>(smx + RR_OFF_B3);
  _Float16* Xl = reinterpret_cast<_Float16*>(smx);
  const int D = r.D, Dst = r.s.c.D;
  if (tid < 256) { b1tab[tid] = nn.W1f[4096 + tid]; b2tab[tid] = nn.b2[tid]; }
  for (int i = tid; i < NA * 256; i += 512) w3tab[i] = i < nn.A * 256 ? nn.W3t[i] : 0.0f;
  for (int i = tid; i < 1024; i += 512) reinterpret_cast<f32x4*>(smx + RR_OFF_W1)[i] = reinterpret_cast<const f32x4*>(nn.W1f)[i];
  for (int i = tid; i < RR_MB * 16; i += 512) { const int e = i >> 4, c = i & 15; obsl[i] = c < D ? r.s.cur_obs[(size_t)(m0 + e) * D + c] : 0.0f; }
  if (tid < 8) b3l[tid] = tid < nn.A ? nn.b3[tid] : 0.0f;
  if (tid < RR_MB) {                                                    // the envs' running state: global -> LDS for the launch
    const int e = m0 + tid;
    float* q = stl + tid * 8;
    q[0] = __int_as_float((int)r.s.next_done[e]); q[1] = r.s.ep_return[e]; q[2] = __int_as_float(r.s.ep_length[e]); q[3] = __int_as_float(r.s.env_t[e]);
    for (int i = 0; i < 4; ++i) q[4 + i] = i < Dst ? r.s.env_state[(size_t)Dst * e + i] : 0.0f;
  }
  RsWeights W;
  rs_load_weights(nn.Wx2, wave, lane, W);
  const float w1un = nn.w1sc[1];
  const float cs = nn.wsc[1] * (1.0f / X2_ACT_SCALE);
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  const int nsteps = r.s.c.k;
  __syncthreads();
  // layer 1 of one 32-env half: observation fragment from LDS, product, activation, split, h1 tile
  auto p_begin = [&](int ct, P2& xb, float& xinv) {
    float xr[8];
    const float* xo = obsl + (32 * ct + j) * 16 + 8 * hf;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(xo), q1 = *reinterpret_cast<const f32x4*>(xo + 4);
    xr[0] = q0[0]; xr[1] = q0[1]; xr[2] = q0[2]; xr[3] = q0[3]; xr[4] = q1[0]; xr[5] = q1[1]; xr[6] = q1[2]; xr[7] = q1[3];
    rs_xfrag<DP>(xr, D, hf, xb, xinv, w1un);
  };
#pragma unroll 1
  for (int step = 0; step < nsteps; ++step) {
    asm volatile("" : "+v"(j), "+v"(hf));                               // per-lane addresses are formed per step, not kept across the loop
    const bool st_ = step == 64;
    if (st_) RS_STAMP(0);
    const f16x8* w1l = reinterpret_cast<const f16x8*>(smx + RR_OFF_W1) + (wave * 2) * 64 + 32 * hf + j;
    // P(0): this wave's 32 units of h1 for envs 0-31
    P2 xb; float xinv;
    p_begin(0, xb, xinv);
    f32x16 c1 = rs_layer1(w1l, xb);
    rs_produce<true, RR_MB>(c1, xinv, b1tab, Xl, wave, j, hf);
    p_begin(1, xb, xinv);
    c1 = rs_layer1(w1l, xb);                                            // the second half's layer-1 product: ahead of the 48 MFMAs it would queue behind
    if (st_) RS_STAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // B1: h1 of envs 0-31 is complete
    if (st_) RS_STAMP(2);
    f32x16 a0, c0;
    // M(0) beside P(1): the matrix pipe multiplies the first half while the vector pipe makes the second half's h1. Four groups of four k-steps, each
    // followed by the activation of four elements; the scheduler may reorder inside a group, not across (one bias quad in flight ahead)
    {
#pragma unroll
      for (int q = 0; q < 16; ++q) { a0[q] = 0.0f; c0[q] = 0.0f; }
      const _Float16* x0 = Xl + j * RS_ROW + 8 * hf;
      const float* b1l = b1tab + 32 * wave + 4 * hf;
      f32x4 bv = *reinterpret_cast<const f32x4*>(b1l);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bn = *reinterpret_cast<const f32x4*>(b1l + 8 * (g < 3 ? g + 1 : 3));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ks = 4 * g + kk;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(x0 + 16 * ks), bl = *reinterpret_cast<const f16x8*>(x0 + RR_MB * RS_ROW + 16 * ks);
          a0 = mfma_f16(W.wl[ks], bh, a0);
          c0 = mfma_f16(W.wh[ks], bh, c0);
          a0 = mfma_f16(W.wh[ks], bl, a0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) c1[4 * g + e] = tanh_fast(__builtin_fmaf(c1[4 * g + e], xinv, bv[e]) * INV_TWO_LOG2E) * X2_ACT_SCALE;   // P(1), in place
        uint2 hh, ll;
        f32x4 hv; hv[0] = c1[4 * g]; hv[1] = c1[4 * g + 1]; hv[2] = c1[4 * g + 2]; hv[3] = c1[4 * g + 3];
        split2x4(hv, 1.0f, hh, ll);
        _Float16* xo = Xl + (32 + j) * RS_ROW + 32 * wave + 4 * hf + 8 * g;
        *reinterpret_cast<uint2*>(xo) = hh;
        *reinterpret_cast<uint2*>(xo + RR_MB * RS_ROW) = ll;
        bv = bn;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) a0[q] += c0[q];
    }
    if (st_) RS_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // B2: h1 of envs 32-63 is complete
    if (st_) RS_STAMP(4);
    // M(1) beside E(0)
    f32x16 a1, c1b;
    {
#pragma unroll
      for (int q = 0; q < 16; ++q) { a1[q] = 0.0f; c1b[q] = 0.0f; }
      const _Float16* x1 = Xl + (32 + j) * RS_ROW + 8 * hf;
      const float* b2l = b2tab + 32 * wave + 4 * hf;
      f32x4 bv = *reinterpret_cast<const f32x4*>(b2l);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bn = *reinterpret_cast<const f32x4*>(b2l + 8 * (g < 3 ? g + 1 : 3));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ks = 4 * g + kk;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(x1 + 16 * ks), bl = *reinterpret_cast<const f16x8*>(x1 + RR_MB * RS_ROW + 16 * ks);
          a1 = mfma_f16(W.wl[ks], bh, a1);
          c1b = mfma_f16(W.wh[ks], bh, c1b);
          a1 = mfma_f16(W.wh[ks], bl, a1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) a0[4 * g + e] = tanh_fast(__builtin_fmaf(a0[4 * g + e], cs, bv[e]));                                  // E(0), in place
        bv = bn;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float* hp = hpall + wave * (RR_MB * 8);
    rs_heads<NA>(a0, w3tab, hp, wave, j, hf);
#pragma unroll
    for (int q = 0; q < 16; ++q) a1[q] += c1b[q];
    rs_epilogue<true, NA>(a1, cs, b2tab, w3tab, hp + 32 * 8, nullptr, wave, j, hf);
    if (st_) RS_STAMP(5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // B3: all head partials are in LDS
    if (st_) RS_STAMP(6);
    if (wave == 0) {
      float z[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) z[q] = 0.0f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {                                  // the eight row groups' partials in fixed order, then the bias
        const float* hq = hpall + w8 * (RR_MB * 8) + lane * 8;
#pragma unroll
        for (int q4 = 0; q4 < NA / 4; ++q4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(hq + 4 * q4);
          z[4 * q4] += v[0]; z[4 * q4 + 1] += v[1]; z[4 * q4 + 2] += v[2]; z[4 * q4 + 3] += v[3];
        }
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) z[q] += b3l[q];
      if (st_) RS_STAMP(8);
      rs_step_env<NA>(r.s, m0 + lane, step, z, obsl + lane * 16, stl + lane * 8, st_n, st_ret, st_len, st_max);
    }
    if (st_) RS_STAMP(7);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // B4: the next observations are in LDS
    if (st_) RS_STAMP(9);
  }
  if (wave == 0) wide_step_stats(r.s.ep_stats, st_n, st_ret, st_len, st_max);
  if (tid < RR_MB) {                                                    // the envs' state back to global memory for the next launch
    const int e = m0 + tid;
    const float* q = stl + tid * 8;
    r.s.next_done[e] = (uint8_t)__float_as_int(q[0]); r.s.ep_return[e] = q[1]; r.s.ep_length[e] = __float_as_int(q[2]); r.s.env_t[e] = __float_as_int(q[3]);
    const bool cart = r.s.c.env_kind == CRL_ENV_CARTPOLE;
    for (int i = 0; i < Dst; ++i) {
      const float o = obsl[tid * 16 + (i < 16 ? i : 0)];
      r.s.cur_obs[(size_t)Dst * e + i] = o;
      r.s.env_state[(size_t)Dst * e + i] = cart ? (i < 4 ? q[4 + i] : 0.0f) : o;   // the synthetic env's state is its observation
    }
  }
}

}  // namespace crl

#ifdef CRL_EXP_WSTAMPS
extern "C" int32_t crl_debug_read_rs_stamps(unsigned long long* out, int32_t n) {
  if (n > 256 * 8 * 16) n = 256 * 8 * 16;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(crl::crl_dbg_rs_stamps), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
