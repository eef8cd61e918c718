// common.hpp — shared device helpers for the gfx950 (MI355X) PPO kernels.
// Wave = 64 lanes. Dense layers run on v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace crl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;        // hidden width (networks.jl:36 default [64,64])
constexpr int TILE = 32;     // samples (or envs) per wave tile = N of the 32x32x2 MFMA

// C/D fragment map of v_mfma_f32_32x32x2_f32: register r of lane l holds D[rowmap(r, l>>5)][l&31]
__host__ __device__ __forceinline__ constexpr int rowmap(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// NNlib tanh_fast (networks.jl:6): rational approximation; n/d via v_rcp_f32 (1 ulp), ±1 beyond x² >= 66
__device__ __forceinline__ float tanh_fast(float x) {
  // the reference switches to sign(x) at x² >= 66; clamping |x| to sqrt(66) instead (one v_med3_f32) returns
  // 0.9999999 there — 1 ulp below the reference's 1.0 — and saves the compare/copysign/select per activation
  x = __builtin_amdgcn_fmed3f(x, -8.1240384f, 8.1240384f);
  float x2 = x * x;
  float n = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 1.587199e-8f, 2.2332108e-5f), 0.0035974074f), 0.1346604f), 1.0f);
  float d = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 8.7767893e-7f, 0.0003453992f), 0.026262015f), 0.4679937f), 1.0f);
  return x * (n * __builtin_amdgcn_rcpf(d));   // v_rcp_f32: 1 ulp
}

// Philox4x32-10 (counter-based; identical stream in oracle/ppo_oracle.c for end-to-end parity runs)
struct u32x4 { uint32_t x, y, z, w; };
__device__ __forceinline__ u32x4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
__device__ __forceinline__ u32x4 philox_env(uint64_t seed, uint32_t gid, uint64_t gstep, uint32_t stream) {
  return philox(gid, (uint32_t)gstep, (uint32_t)(gstep >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ double u53(u32x4 o) {
  uint64_t bits = (((uint64_t)o.x << 32) | o.y) >> 11;
  return (double)bits * 0x1.0p-53;
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
// v + xor32(v) and max(v, xor32(v)) without the LDS crossbar: v_permlane32_swap_b32 (gfx950) swaps the upper half of its first operand with the lower half of
// its second, so two copies of v come back as {lower, lower} and {upper, upper} — a vector instruction, where __shfl_xor is a ds_bpermute_b32 with an lgkmcnt
// round trip. Same bits as the shuffle form (lower + upper in both halves). Inline asm: with __builtin_amdgcn_permlane32_swap this compiler (ROCm 7.2) hands out
// the FIRST result for both elements of the returned pair (r[0] + r[1] compiles to 2·r[0]).
__device__ __forceinline__ void swap32(float v, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a); hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float add32(float v) { float lo, hi; swap32(v, lo, hi); return lo + hi; }
__device__ __forceinline__ float max32(float v) { float lo, hi; swap32(v, lo, hi); return __builtin_fmaxf(lo, hi); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Orders one wave's LDS traffic between phases of wave-private scratch use (no s_barrier needed: the LDS
// pipeline executes a wave's DS instructions in issue order; this only pins the compiler's order).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------
// Per-network weight image in LDS (floats). Built by stage_net() from the flat Flux-ordered parameters.
//   wf2[mo][s4][lane][4]   forward hidden layer A-fragments:  W2[32mo+i][32mt+rowmap(r,hf)],  s = 16mt+r
//   wb2[mo][s4][lane][4]   backward (W2^T) A-fragments:       W2[32mt+rowmap(r,hf)][32mo+i]
//   wf1[mo][ks][lane]      first layer A-fragments:           W1[32mo+i][2ks+hf]
//   b1c/b2c[hf][mo][16]    biases in C-fragment order:        b[32mo+rowmap(r,hf)]
//   w3[a][hf][32]          head rows in C-fragment order:     W3[a][32mt+rowmap(r,hf)], index 16mt+r
//   b3[a]
// ------------------------------------------------------------------------------------------------------
template <int D, int NOUT, bool BWD>
struct NetImage {
  static constexpr int WF2 = 0;
  static constexpr int WB2 = WF2 + 4096;
  static constexpr int WF1 = WB2 + (BWD ? 4096 : 0);
  static constexpr int B1C = WF1 + 2 * (D / 2) * 64;
  static constexpr int B2C = B1C + 64;
  static constexpr int W3 = B2C + 64;
  static constexpr int B3 = W3 + NOUT * 64;
  static constexpr int SIZE = ((B3 + NOUT + 3) / 4) * 4;
};

// Flat parameter offsets of one network (actor: base 0; critic: base = actor size)
template <int D, int NOUT>
struct NetParams {
  static constexpr int W1 = 0;
  static constexpr int B1 = W1 + H * D;
  static constexpr int W2 = B1 + H;
  static constexpr int B2 = W2 + H * H;
  static constexpr int W3 = B2 + H;
  static constexpr int B3 = W3 + NOUT * H;
  static constexpr int SIZE = B3 + NOUT;
};

template <int D, int NOUT, bool BWD>
__device__ __forceinline__ void stage_net(float* img, const float* __restrict__ p, int tid, int nthreads) {
  using I = NetImage<D, NOUT, BWD>;
  using P = NetParams<D, NOUT>;
  for (int idx = tid; idx < 4096; idx += nthreads) {
    int e = idx & 3, lane = (idx >> 2) & 63, s4 = (idx >> 8) & 7, mo = idx >> 11;
    int s = s4 * 4 + e, mt = s >> 4, r = s & 15, i = lane & 31, hf = lane >> 5;
    int row = 32 * mo + i, k = 32 * mt + rowmap(r, hf);
    img[I::WF2 + idx] = p[P::W2 + row + H * k];
    if (BWD) img[I::WB2 + idx] = p[P::W2 + k + H * row];
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

__device__ __forceinline__ f32x16 load16(const float* p) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
  f32x16 o;
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  o[8] = c[0]; o[9] = c[1]; o[10] = c[2]; o[11] = c[3]; o[12] = d[0]; o[13] = d[1]; o[14] = d[2]; o[15] = d[3];
  return o;
}

// Forward of one network for a 32-sample tile. x[D] is this lane's sample (lane&31); both lane halves hold the
// same sample. Outputs h1/h2 in C-fragment layout (kept for the backward pass) and the NOUT head outputs.
template <int D, int NOUT, bool BWD>
__device__ __forceinline__ void mlp_forward(const float* img, const float (&x)[D], f32x16 (&h1)[2], f32x16 (&h2)[2],
                                            float (&out)[NOUT], int lane) {
  using I = NetImage<D, NOUT, BWD>;
  const int hf = lane >> 5;
  // layer 1: K = D, two rows of x per k-step (lane half hf supplies x[2ks+hf])
  f32x16 a0 = load16(img + I::B1C + hf * 32);
  f32x16 a1 = load16(img + I::B1C + hf * 32 + 16);
#pragma unroll
  for (int ks = 0; ks < D / 2; ++ks) {
    float b = hf ? x[2 * ks + 1] : x[2 * ks];
    a0 = mfma32(img[I::WF1 + (0 * (D / 2) + ks) * 64 + lane], b, a0);
    a1 = mfma32(img[I::WF1 + (1 * (D / 2) + ks) * 64 + lane], b, a1);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { h1[0][r] = tanh_fast(a0[r]); h1[1][r] = tanh_fast(a1[r]); }
  // layer 2: 32 k-steps, B operand straight from the C-fragment registers of h1
  a0 = load16(img + I::B2C + hf * 32);
  a1 = load16(img + I::B2C + hf * 32 + 16);
  const f32x4* w0 = reinterpret_cast<const f32x4*>(img + I::WF2) + lane;
  const f32x4* w1 = reinterpret_cast<const f32x4*>(img + I::WF2 + 2048) + lane;
#pragma unroll
  for (int s4 = 0; s4 < 8; ++s4) {
    f32x4 fa = w0[s4 * 64], fb = w1[s4 * 64];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = s4 * 4 + e;
      float b = h1[s >> 4][s & 15];
      a0 = mfma32(fa[e], b, a0);
      a1 = mfma32(fb[e], b, a1);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { h2[0][r] = tanh_fast(a0[r]); h2[1][r] = tanh_fast(a1[r]); }
  // head: each lane half sums its 32 rows, halves combine through one cross-half exchange
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

// softmax / logsoftmax of NNlib (ppo.jl:23-24,36-37) on A logits held in registers
template <int A>
__device__ __forceinline__ void softmax_logsoftmax(const float (&z)[A], float (&p)[A], float (&lp)[A]) {
  float m = z[0];
#pragma unroll
  for (int a = 1; a < A; ++a) m = fmaxf(m, z[a]);
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) { p[a] = expf(z[a] - m); s += p[a]; }
#pragma unroll
  for (int a = 0; a < A; ++a) p[a] = p[a] / s;
  float ls = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) { lp[a] = z[a] - m; ls += expf(lp[a]); }
  float l = logf(ls);
#pragma unroll
  for (int a = 0; a < A; ++a) lp[a] = lp[a] - l;
}

// StatsBase.sample(Weights(p)) (ppo.jl:26): f32 running sum against the f64 threshold u*sum(p)
template <int A>
__device__ __forceinline__ int sample_weights(const float (&p)[A], double u) {
  float sw = 0.0f;
#pragma unroll
  for (int a = 0; a < A; ++a) sw += p[a];
  double t = u * (double)sw;
  int i = 0;
  float cw = p[0];
#pragma unroll
  for (int a = 1; a < A; ++a) {
    bool go = ((double)cw < t) && (i == a - 1);
    i = go ? a : i;
    cw = go ? cw + p[a] : cw;
  }
  return i;
}

}  // namespace crl
