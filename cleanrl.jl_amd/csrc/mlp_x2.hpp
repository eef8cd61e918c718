// mlp_x2.hpp — the 64x64 dense layers on the f16 matrix pipe at float32 accuracy ("fp16x2").
//
// mlp_x3.hpp splits every f32 operand into three bf16 pieces and spends six MFMAs per product; the ablation runs of the
// update kernel show its GEMM phases bound by exactly that matrix-pipe time. A half-precision float has 11
// significant bits, so TWO pieces x = hi + lo carry 22 bits and a product needs three MFMAs (hi·hi, hi·lo, lo·hi; the
// dropped lo·lo term is 2^-22 relative) — half the matrix-pipe time, and the split costs 2 VALU instructions per element
// instead of 5.5 (the residual x − hi is one mixed-precision subtract). The price is fp16's 5-bit exponent: a piece pair keeps
// its 22 bits only while |x| stays in [2^-3, 65504); below that the absolute error floor is 2^-25. So every operand is scaled
// by an exact power of two into that window, and the scale comes back out of the f32 accumulator:
//   * activations h = tanh(·) ∈ (−1, 1):  h·2^14, produced directly by the activation (tanh_exp2 with S = 2^14: the scale is
//     the constant of its last fused multiply-add);
//   * weights:  W·2^8 — full precision for |w| ≥ 2^-11, absolute error 2^-33 below; |w| ≥ 255 does not fit: the block that finds
//     one while staging runs that network as bf16x3 for the launch (update.hip, policy.hip; option gemm = 1 selects bf16x3 everywhere);
//   * backward cotangents δ: any magnitude — each SAMPLE (= lane: the N index of the product) is scaled by its own power of
//     two, taken from the largest |δ| of that sample, and unscaled after the product (a per-column scale commutes with A·B).
// The weight-gradient product sums over samples (K = samples), where a per-sample scale does not commute: it takes ONE scale
// per launch, predicted from the previous launch and checked per tile, with bf16x3 as the in-kernel fallback (end of file).
#pragma once
#include "common.hpp"
#include "mlp_x3.hpp"

namespace crl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr float X2_ACT_SCALE = 16384.0f;          // 2^14 on tanh outputs
constexpr float X2_W_SCALE = 256.0f;              // 2^8 on weights
constexpr float X2_W_LIMIT = 255.0f;              // |w|·2^8 must stay below the largest half (65504)
constexpr float X2_FWD_UNSCALE = 1.0f / (16384.0f * 256.0f);

struct P2 { f16x8 hi, lo; };

// x = hi + lo for 8 floats (already scaled into the fp16 window): per pair one v_cvt_pk_f16_f32, two mixed-precision
// subtracts (x − float(hi), exact) and one more v_cvt_pk_f16_f32
__device__ __forceinline__ P2 split2(const float (&x)[8]) {
  P2 p;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x2 v; v[0] = x[2 * q]; v[1] = x[2 * q + 1];
    const f16x2 h = __builtin_convertvector(v, f16x2);
    // x − float(hi) = fma(hi, −1, x) as ONE v_fma_mix_f32 (the half operand is widened inside the instruction; written as asm
    // because the compiler canonicalises the expression to v_cvt_f32_f16 + v_sub_f32)
    f32x2 r;
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hb), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hb), "v"(v[1]));
    const f16x2 l = __builtin_convertvector(r, f16x2);
    p.hi[2 * q] = h[0]; p.hi[2 * q + 1] = h[1];
    p.lo[2 * q] = l[0]; p.lo[2 * q + 1] = l[1];
  }
  return p;
}

__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// smallest partial products first
__device__ __forceinline__ f32x16 mfma_x2(const P2& a, const P2& b, f32x16 c) {
  c = mfma_f16(a.lo, b.hi, c);
  c = mfma_f16(a.hi, b.lo, c);
  c = mfma_f16(a.hi, b.hi, c);
  return c;
}

// ------------------------------------------------------------------------------------------------------
// LDS weight image (offsets in floats; a piece image is 4096 halves = 2048 floats), same fragment order as NetImageX3:
//   wf2h[piece][mo][ks][lane][8]  forward A-fragments of W2·2^8, wb2h … of W2ᵀ·2^8; k order = kmap (mlp_x3.hpp)
//   wf1 / b1c / b2c / w3 / b3 as in NetImage (f32)
// ------------------------------------------------------------------------------------------------------
// BWD = false (the rollout's critic: forward only) leaves the W2ᵀ pieces out: 16 KB less LDS per block, which is what lets a second
// shuffle block share a CU with a rollout block inside crl_ppo_iterate
template <int D, int NOUT, bool BWD = true>
struct NetImageX2 {
  static constexpr int PIECE = 2048;
  static constexpr int WF2H = 0;
  static constexpr int WB2H = WF2H + 2 * PIECE;
  static constexpr int WF1 = WB2H + (BWD ? 2 * PIECE : 0);
  static constexpr int B1C = WF1 + 2 * (D / 2) * 64;
  static constexpr int B2C = B1C + 64;
  static constexpr int W3 = B2C + 64;
  static constexpr int B3 = W3 + NOUT * 64;
  static constexpr int SIZE = ((B3 + NOUT + 3) / 4) * 4;
};

// returns false (for every thread of the block) when a weight of the hidden layer does not fit the fp16 window
template <int D, int NOUT, bool BWD = true>
__device__ __forceinline__ bool stage_net_x2(float* img, const float* __restrict__ p, int tid, int nthreads, int* lds_flag) {
  using I = NetImageX2<D, NOUT, BWD>;
  using P = NetParams<D, NOUT>;
  _Float16* wf = reinterpret_cast<_Float16*>(img + I::WF2H);
  _Float16* wb = reinterpret_cast<_Float16*>(img + I::WB2H);
  if (tid == 0) *lds_flag = 0;
  __syncthreads();
  bool bad = false;
  for (int idx = tid; idx < 4096; idx += nthreads) {
    const int j = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 3, mo = idx >> 11;
    const int i = lane & 31, hf = lane >> 5;
    const int row = 32 * mo + i, k = kmap(ks, j, hf);
    {
      const float w0 = p[P::W2 + row + H * k];
      bad |= !(__builtin_fabsf(w0) < X2_W_LIMIT);
      const float w = w0 * X2_W_SCALE;
      const _Float16 h = (_Float16)w;
      wf[idx] = h; wf[4096 + idx] = (_Float16)(w - (float)h);
    }
    if (BWD) {
      const float w = p[P::W2 + k + H * row] * X2_W_SCALE;
      const _Float16 h = (_Float16)w;
      wb[idx] = h; wb[4096 + idx] = (_Float16)(w - (float)h);
    }
  }
  for (int idx = tid; idx < 2 * (D / 2) * 64; idx += nthreads) {
    int lane = idx & 63, ks = (idx >> 6) % (D / 2), mo = (idx >> 6) / (D / 2);
    int i = lane & 31, hf = lane >> 5;
    // layer 1 is staged ×2·log2(e) (bias too): its accumulator is the exponent tanh_exp2_arg wants, no multiply per element
    img[I::WF1 + idx] = p[P::W1 + (32 * mo + i) + H * (2 * ks + hf)] * TWO_LOG2E;
  }
  for (int idx = tid; idx < 64; idx += nthreads) {
    int r = idx & 15, mo = (idx >> 4) & 1, hf = idx >> 5;
    int row = 32 * mo + rowmap(r, hf);
    img[I::B1C + idx] = p[P::B1 + row] * TWO_LOG2E;
    img[I::B2C + idx] = p[P::B2 + row] * (X2_ACT_SCALE * X2_W_SCALE);   // the layer-2 accumulator starts at b2·2^22
  }
  for (int idx = tid; idx < NOUT * 64; idx += nthreads) {
    int q = idx & 31, hf = (idx >> 5) & 1, a = idx >> 6;
    int mt = q >> 4, r = q & 15;
    img[I::W3 + idx] = p[P::W3 + a + NOUT * (32 * mt + rowmap(r, hf))];
  }
  for (int idx = tid; idx < NOUT; idx += nthreads) img[I::B3 + idx] = p[P::B3 + idx];
  if (bad) *lds_flag = 1;
  __syncthreads();
  return *lds_flag == 0;
}

__device__ __forceinline__ P2 load_wfrag2(const float* piece0, int mo, int ks, int lane) {
  const f16x8* q = reinterpret_cast<const f16x8*>(piece0) + ((mo * 4 + ks) * 64 + lane);
  P2 a;
  a.hi = q[0]; a.lo = q[512];   // pieces are 4096 halves = 512 fragments apart
  return a;
}

// acc[mo] += (W·2^8)(64x64) · Xs(64 x 32 samples), Xs already scaled into the fp16 window, given as C-fragment registers
__device__ __forceinline__ void dense64_x2(const float* wimg, const f32x16 (&xs)[2], f32x16& acc0, f32x16& acc1, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    float xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = xs[ks >> 1][8 * (ks & 1) + j];
    const P2 b = split2(xb);
    acc0 = mfma_x2(load_wfrag2(wimg, 0, ks, lane), b, acc0);
    acc1 = mfma_x2(load_wfrag2(wimg, 1, ks, lane), b, acc1);
  }
}

// Forward of one network for a 32-sample tile: h1s = 2^14·h1 (what the next product and the backward pass consume), h2 and
// the head outputs unscaled
template <int D, int NOUT, bool BWD = true>
__device__ __forceinline__ void mlp_forward_x2(const float* img, const float (&x)[D], f32x16 (&h1s)[2], f32x16 (&h2)[2],
                                               float (&out)[NOUT], int lane) {
  using I = NetImageX2<D, NOUT, BWD>;
  const int hf = lane >> 5;
  f32x16 a0 = load16(img + I::B1C + hf * 32);
  f32x16 a1 = load16(img + I::B1C + hf * 32 + 16);
#pragma unroll
  for (int ks = 0; ks < D / 2; ++ks) {
    const float b = hf ? x[2 * ks + 1] : x[2 * ks];
    a0 = mfma32(img[I::WF1 + (0 * (D / 2) + ks) * 64 + lane], b, a0);
    a1 = mfma32(img[I::WF1 + (1 * (D / 2) + ks) * 64 + lane], b, a1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { h1s[0][r] = tanh_exp2_arg(a0[r], X2_ACT_SCALE); h1s[1][r] = tanh_exp2_arg(a1[r], X2_ACT_SCALE); }   // W1, b1 staged ×2·log2(e)
  a0 = load16(img + I::B2C + hf * 32);
  a1 = load16(img + I::B2C + hf * 32 + 16);
  dense64_x2(img + I::WF2H, h1s, a0, a1, lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) {   // the fp16x2 unscale rides in the exponent's multiplier
    h2[0][r] = tanh_exp2(a0[r], TWO_LOG2E * X2_FWD_UNSCALE, 1.0f); h2[1][r] = tanh_exp2(a1[r], TWO_LOG2E * X2_FWD_UNSCALE, 1.0f);
  }
#pragma unroll
  for (int a = 0; a < NOUT; ++a) {
    const f32x4* w = reinterpret_cast<const f32x4*>(img + I::W3 + a * 64 + hf * 32);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 wv = w[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = q * 4 + e;
        acc = __builtin_fmaf(wv[e], h2[idx >> 4][idx & 15], acc);
      }
    }
    out[a] = add32(acc) + img[I::B3 + a];
  }
}

// Per-sample power-of-two scale for a cotangent tile in C-fragment registers (lane = sample, both lane halves hold rows of
// the same sample): s = 2^(14 − ⌈exponent of the sample's largest |δ|⌉), exact; inv = 1 / s; m = that largest |δ|.
__device__ __forceinline__ void sample_scale(const f32x16 (&d)[2], float& s, float& inv, float& m) {
  m = 0.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(d[mt][r]), __builtin_fabsf(d[mt][r + 1])));
  m = max32(m);
  int e = (int)((__float_as_uint(m) >> 23) & 0xFFu);   // biased exponent: m in [2^(e-127), 2^(e-126))
  e = e < 16 ? 16 : e;                                   // zero / tiny columns: any scale will do
  s = __uint_as_float((unsigned)(268 - e) << 23);        // 2^(141 − e): m·s in [2^14, 2^15)
  inv = __uint_as_float((unsigned)(e - 14) << 23);       // 2^(e − 141)
}

// ------------------------------------------------------------------------------------------------------
// Weight-gradient product dW2ᵀ += h1·δ2ᵀ (K = samples) on fp16x2. A per-sample scale does not commute with a sum over
// samples, and the accumulators live across all tiles of a launch, so δ2 gets ONE power of two G per launch and role. G is
// carried from launch to launch on the device: every launch records the largest |δ2| it saw, the reduce kernel turns it
// into the next launch's G (largest·G ≈ 2^8, exponent quantised to multiples of 8 so that it rarely changes). G is a
// prediction, so every tile checks it: if one of its samples would overflow (|δ2|·G ≥ 2^15.5) or all of them sit below the
// window (|δ2|·G < 2^-6) the tile takes the bf16x3 path instead — same operands, same scale, no range limits — a
// wave-uniform branch. Nothing depends on the prediction being right; a wrong one only costs speed.
// ------------------------------------------------------------------------------------------------------
constexpr float X2_DW_OVER = 46340.0f;        // ≈ 2^15.5
constexpr float X2_DW_SMALL = 0.015625f;      // 2^-6

__device__ __forceinline__ bool dw_tile_fits(float m, float G) {
  const float v = m * G;
  const bool over = !(v < X2_DW_OVER);        // also catches NaN
  const bool notsmall = v >= X2_DW_SMALL;
  return __builtin_amdgcn_ballot_w64(over) == 0 && __builtin_amdgcn_ballot_w64(notsmall) != 0;
}

// next launch's G from the largest |δ2| of this one (bits of a non-negative float order like unsigned integers). G is
// sticky: it stays as long as largest·G sits in [2^0, 2^13] and is re-centred (largest·G ≈ 2^8, exponent a multiple of 8)
// only when the data have moved out of that band — so consecutive launches on similar data use the same G and the same
// inputs give bit-identical gradients (a different G changes which elements touch fp16's subnormal floor).
__device__ __forceinline__ float dw_next_scale(unsigned max_bits, float G_old) {
  if (max_bits == 0) return G_old;
  const float v = __uint_as_float(max_bits) * G_old;
  if (v >= 1.0f && v <= 8192.0f) return G_old;
  const int e = (int)((max_bits >> 23) & 0xFFu) - 127;    // largest in [2^e, 2^(e+1))
  int k = 8 - e;                                           // largest·2^k in [2^8, 2^9)
  k = (k >= 0 ? (k + 4) / 8 : -((-k + 4) / 8)) * 8;        // exponent quantised to a multiple of 8
  k = k < -100 ? -100 : (k > 100 ? 100 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}

}  // namespace crl
