// mlp_x3.hpp — the 64x64 dense layers on the bf16 matrix pipe at float32 accuracy ("bf16x3").
//
// Measured on MI355X (profiles/r01_b_pmc_summary.json): v_mfma_f32_32x32x2_f32 NEVER co-executes with VALU work
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0) — the f32 matrix instruction runs at the vector rate and time adds up as
// MFMA + VALU. The bf16 matrix pipe is separate (16x the f32 rate) and does overlap VALU. So each f32 operand is split
// into three bf16 pieces x = hi + mid + lo (round-to-nearest residuals, 8+8+8 significand bits) and a product keeps the six
// partial products down to 2^-16 relative weight:  hi·hi, hi·mid, mid·hi, mid·mid, hi·lo, lo·hi.  Every bf16×bf16 product is
// exact in f32 and the accumulator is f32, so the dropped terms (≤ 2^-24 relative) are at the level of f32 rounding itself.
// Cost per 64x64x32 layer: 48 v_mfma_f32_32x32x16_bf16 (1,536 matrix-pipe cycles, overlappable) instead of 64
// v_mfma_f32_32x32x2_f32 (4,096 cycles that block the VALU), plus ~5.5 VALU ops per activation element for the split.
#pragma once
#include "common.hpp"

namespace crl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float TWO_LOG2E = 2.8853900817779268f;  // 2·log2(e): tanh(x) = 1 − 2/(2^(TWO_LOG2E·x) + 1)
// The activation of the update kernels (fp16x2, and since round 6 update_x3_kernel too — the operand width of the products is what `gemm = 1` is about, not the
// tanh approximation; the rollout's actor keeps the reference's rational tanh_fast because its logits decide action indices, the bf16x3 fallbacks because they run
// when weights are extreme): S·tanh(x·c) = S − 2S / (2^(x·pre) + 1) with pre = 2·log2(e)·c — five
// instructions (v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma) instead of the thirteen of the rational tanh_fast (networks.jl:6),
// which were 40 % of the kernel's VALU instructions (the kernel is VALU-issue-bound: §3 of DESIGN.md). Both approximate tanh:
// the rational form to a few ulp relative, this one to ≈1e-7 ABSOLUTE (v_exp_f32 / v_rcp_f32 are 1 ulp; 1 − 2r cancels for
// small |x|), which is the rounding unit of an activation in (−1, 1) anyway. Measured against the oracle (which evaluates the
// reference's rational form): gradients 1.16e-6 vs 1.12e-6 relative L2, parameters after three iterations 3e-8 vs 2e-8
// (profiles/r02_parity_margins.json) — far inside the 1e-5 bar. Saturation needs no clamp: 2^(+big) = inf → S, 2^(−big) = 0 → −S.
// The rollout keeps tanh_fast: action indices there are bit-compared with the oracle.
__device__ __forceinline__ float tanh_exp2_arg(float t, float S) {   // t = 2·log2(e)·x already
  const float e = __builtin_amdgcn_exp2f(t);
  const float r = __builtin_amdgcn_rcpf(e + 1.0f);
  return __builtin_fmaf(-2.0f * S, r, S);
}
__device__ __forceinline__ float tanh_exp2(float x, float pre, float S) { return tanh_exp2_arg(x * pre, S); }


struct P3 { bf16x8 hi, mid, lo; };

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// one v_cvt_pk_bf16_f32: two floats → packed bf16 pair (round to nearest even)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  f32x2 v; v[0] = a; v[1] = b;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// x = hi + mid + lo for 8 floats: per pair 3 converts + 4 bit ops + 4 subtracts
__device__ __forceinline__ P3 split3(const float (&x)[8]) {
  u32x4v h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = x[2 * q], x1 = x[2 * q + 1];
    const uint32_t hp = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);
    const uint32_t mp = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);
    h[q] = hp; m[q] = mp; l[q] = cvt_pk_bf16(s0, s1);
  }
  P3 p;
  p.hi = __builtin_bit_cast(bf16x8, h); p.mid = __builtin_bit_cast(bf16x8, m); p.lo = __builtin_bit_cast(bf16x8, l);
  return p;
}

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// smallest partial products first
__device__ __forceinline__ f32x16 mfma_x3(const P3& a, const P3& b, f32x16 c) {
  c = mfma_bf16(a.lo, b.hi, c);
  c = mfma_bf16(a.hi, b.lo, c);
  c = mfma_bf16(a.mid, b.mid, c);
  c = mfma_bf16(a.mid, b.hi, c);
  c = mfma_bf16(a.hi, b.mid, c);
  c = mfma_bf16(a.hi, b.hi, c);
  return c;
}

// ------------------------------------------------------------------------------------------------------
// LDS weight image for the bf16x3 path (offsets in floats; the bf16 arrays take half a float per element).
//   wf2p[piece][mo][ks][lane][8]  forward A-fragments:  W2[32mo+i][kmap(ks,j,hf)]      (k-step ks = 2mt+s covers 16 k)
//   wb2p[piece][mo][ks][lane][8]  backward (W2^T):      W2[kmap(ks,j,hf)][32mo+i]
//   kmap(ks, j, hf) = 32mt + rowmap(8s+j, hf): the k order in which the C-fragment registers 8s..8s+7 of the previous
//   layer hold their rows, so that fragment is the B operand as it stands (no lane movement).
//   wf1 / b1c / b2c / w3 / b3 as in NetImage (f32).
// ------------------------------------------------------------------------------------------------------
template <int D, int NOUT, bool BWD>
struct NetImageX3 {
  static constexpr int PIECE = 2 * 4 * 64 * 8 / 2;  // floats per piece image (4096 bf16)
  static constexpr int WF2P = 0;
  static constexpr int WB2P = WF2P + 3 * PIECE;
  static constexpr int WF1 = WB2P + (BWD ? 3 * PIECE : 0);
  static constexpr int B1C = WF1 + 2 * (D / 2) * 64;
  static constexpr int B2C = B1C + 64;
  static constexpr int W3 = B2C + 64;
  static constexpr int B3 = W3 + NOUT * 64;
  static constexpr int SIZE = ((B3 + NOUT + 3) / 4) * 4;
};

__device__ __forceinline__ int kmap(int ks, int j, int hf) { return 32 * (ks >> 1) + rowmap(8 * (ks & 1) + j, hf); }

template <int D, int NOUT, bool BWD>
__device__ __forceinline__ void stage_net_x3(float* img, const float* __restrict__ p, int tid, int nthreads) {
  using I = NetImageX3<D, NOUT, BWD>;
  using P = NetParams<D, NOUT>;
  __bf16* wf = reinterpret_cast<__bf16*>(img + I::WF2P);
  __bf16* wb = reinterpret_cast<__bf16*>(img + I::WB2P);
  for (int idx = tid; idx < 4096; idx += nthreads) {
    const int j = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) & 3, mo = idx >> 11;
    const int i = lane & 31, hf = lane >> 5;
    const int row = 32 * mo + i, k = kmap(ks, j, hf);
    {
      const float w = p[P::W2 + row + H * k];
      const __bf16 h = (__bf16)w; const float r1 = w - (float)h;
      const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
      wf[idx] = h; wf[4096 + idx] = m; wf[8192 + idx] = (__bf16)r2;
    }
    if (BWD) {
      const float w = p[P::W2 + k + H * row];
      const __bf16 h = (__bf16)w; const float r1 = w - (float)h;
      const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
      wb[idx] = h; wb[4096 + idx] = m; wb[8192 + idx] = (__bf16)r2;
    }
  }
  for (int idx = tid; idx < 2 * (D / 2) * 64; idx += nthreads) {
    int lane = idx & 63, ks = (idx >> 6) % (D / 2), mo = (idx >> 6) / (D / 2);
    int i = lane & 31, hf = lane >> 5;
    img[I::WF1 + idx] = p[P::W1 + (32 * mo + i) + H * (2 * ks + hf)];
  }
  for (int idx = tid; idx < 64; idx += nthreads) {
    int r = idx & 15, mo = (idx >> 4) & 1, hf = idx >> 5;
    int row = 32 * mo + rowmap(r, hf);
    img[I::B1C + idx] = p[P::B1 + row];
    img[I::B2C + idx] = p[P::B2 + row];
  }
  for (int idx = tid; idx < NOUT * 64; idx += nthreads) {
    int q = idx & 31, hf = (idx >> 5) & 1, a = idx >> 6;
    int mt = q >> 4, r = q & 15;
    img[I::W3 + idx] = p[P::W3 + a + NOUT * (32 * mt + rowmap(r, hf))];
  }
  for (int idx = tid; idx < NOUT; idx += nthreads) img[I::B3 + idx] = p[P::B3 + idx];
}

// A-fragment pieces of one (mo, ks) block: three 16-byte LDS reads
__device__ __forceinline__ P3 load_wfrag(const float* piece0, int mo, int ks, int lane) {
  const bf16x8* q = reinterpret_cast<const bf16x8*>(piece0) + ((mo * 4 + ks) * 64 + lane);
  P3 a;
  a.hi = q[0]; a.mid = q[512]; a.lo = q[1024];   // pieces are 4096 bf16 = 512 fragments apart
  return a;
}

// acc[mo] += W(64x64) · X(64 x 32 samples) with X given as C-fragment registers (x[mt][r]); wimg = wf2p or wb2p.
// (Tried and dropped, DESIGN.md §3: software-pipelining the k-steps with sched_group_barrier — next step's fragment reads and
// split under this step's MFMAs — costs 90-150 spilled registers at the 256-register budget and ran 1-6 % slower.)
__device__ __forceinline__ P3 split_kstep(const f32x16 (&x)[2], int ks) {
  float xb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xb[j] = x[ks >> 1][8 * (ks & 1) + j];
  return split3(xb);
}
__device__ __forceinline__ void dense64_x3(const float* wimg, const f32x16 (&x)[2], f32x16& acc0, f32x16& acc1, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const P3 b = split_kstep(x, ks);
    acc0 = mfma_x3(load_wfrag(wimg, 0, ks, lane), b, acc0);
    acc1 = mfma_x3(load_wfrag(wimg, 1, ks, lane), b, acc1);
  }
}

// Forward of one network for a 32-sample tile (same contract as mlp_forward in common.hpp)
// DBG: bit 7 (128) = the exp2-based activation of tanh_exp2 instead of the rational tanh_fast — what update_x3_kernel (option gemm = 1) uses, like the fp16x2
// update pass (5 vector instructions per element instead of 13: 0.81 -> 0.73 ms per launch); timing experiments only (CRL_ABLATE builds): bit 2 = cheap
// activation instead of tanh_fast, bit 6 = no layer-2 MFMAs
template <int D, int NOUT, bool BWD, int DBG = 0>
__device__ __forceinline__ void mlp_forward_x3(const float* img, const float (&x)[D], f32x16 (&h1)[2], f32x16 (&h2)[2],
                                               float (&out)[NOUT], int lane) {
  using I = NetImageX3<D, NOUT, BWD>;
  const int hf = lane >> 5;
  // layer 1 (K = D = 4): two f32 MFMAs per output block — too thin for the bf16 shape
  f32x16 a0 = load16(img + I::B1C + hf * 32);
  f32x16 a1 = load16(img + I::B1C + hf * 32 + 16);
#pragma unroll
  for (int ks = 0; ks < D / 2; ++ks) {
    float b = hf ? x[2 * ks + 1] : x[2 * ks];
    a0 = mfma32(img[I::WF1 + (0 * (D / 2) + ks) * 64 + lane], b, a0);
    a1 = mfma32(img[I::WF1 + (1 * (D / 2) + ks) * 64 + lane], b, a1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (DBG & 4) { h1[0][r] = a0[r] * 0.5f; h1[1][r] = a1[r] * 0.5f; }
    else if constexpr (DBG & 128) { h1[0][r] = tanh_exp2(a0[r], TWO_LOG2E, 1.0f); h1[1][r] = tanh_exp2(a1[r], TWO_LOG2E, 1.0f); }
    else { h1[0][r] = tanh_fast(a0[r]); h1[1][r] = tanh_fast(a1[r]); }
  }
  // layer 2 on the bf16 pipe
  a0 = load16(img + I::B2C + hf * 32);
  a1 = load16(img + I::B2C + hf * 32 + 16);
  if constexpr (DBG & 64) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] += h1[0][r]; a1[r] += h1[1][r]; }
  } else dense64_x3(img + I::WF2P, h1, a0, a1, lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if constexpr (DBG & 4) { h2[0][r] = a0[r] * 0.5f; h2[1][r] = a1[r] * 0.5f; }
    else if constexpr (DBG & 128) { h2[0][r] = tanh_exp2(a0[r], TWO_LOG2E, 1.0f); h2[1][r] = tanh_exp2(a1[r], TWO_LOG2E, 1.0f); }
    else { h2[0][r] = tanh_fast(a0[r]); h2[1][r] = tanh_fast(a1[r]); }
  }
  // head on VALU
#pragma unroll
  for (int a = 0; a < NOUT; ++a) {
    const f32x4* w = reinterpret_cast<const f32x4*>(img + I::W3 + a * 64 + hf * 32);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 wv = w[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int idx = q * 4 + e;
        acc = __builtin_fmaf(wv[e], h2[idx >> 4][idx & 15], acc);
      }
    }
    out[a] = add32(acc) + img[I::B3 + a];
  }
}

}  // namespace crl
