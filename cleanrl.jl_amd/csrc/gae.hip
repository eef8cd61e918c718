// gae.hip — GAE advantages + returns (ppo.jl:48-73,169-181) as a segmented reverse scan.
//
// The recurrence A_t = δ_t + c_t·A_{t+1} (δ_t = r_t + γ·nt_{t+1}·v_{t+1} − v_t, c_t = (γλ)·nt_{t+1}) is affine, so the
// time axis is cut into segments of L=8/16 steps that run in parallel: each thread owns (env, segment), keeps its L
// (v, r, done) triples in registers, composes the segment's map A_lo = D + C·A_in, the per-segment maps are folded
// through LDS, and the thread replays its segment from the resolved carry. One HBM read of value/reward/terminal
// (9 B) and one write of advantage/return (8 B) per (env, step): 17 B — the algorithmic minimum.
// Arithmetic is Float64 like the reference's accumulator (ppo.jl:63,65; Q2), stored Float32 (ppo.jl:62).
// Layout: (nt, k) column-major, env fastest ⇒ lanes of a wave read consecutive envs: coalesced 128/256-B rows.
//
// Where it stands (bench.py roofline_gae, 65536 envs): 29.6 µs on inputs the previous kernel left in cache (0.60 of 8 TB/s); with caches
// flushed 51 µs with cached loads and 41 µs with nontemporal loads (the NTL flavour: 0.43 of 8 TB/s) — exactly what a plain copy of the
// same number of bytes takes under the same cold conditions (41 µs). Below ~4 M samples the streaming loads do not pay (18.5 vs 17.8 µs at
// 16384 envs). Round 3 also measured three rewrites and kept none: four envs per thread with 16-byte accesses and a dword of done flags
// (a quarter of the vector-memory instructions; 128 registers, L = 4): 50-52 µs cold, 35 µs warm; a persistent grid with the next tile's
// inputs double-buffered in registers: 52.6 / 35 µs; tighter register caps (80 registers: spills).
#include <hip/hip_ext.h>

#include <cstdlib>

#include "ppo_ctx.hpp"

namespace crl {

// NTL: nontemporal loads of the three input streams — for inputs that are NOT in the caches (host-driven rollouts whose buffers arrive
// by copies; bench.py's flushed-cache measurement): 41 µs instead of 50 at 65536 x 128, the time of a plain copy of the same bytes.
// Behind an on-device rollout the inputs sit in L2 / Infinity Cache and the cached loads are the faster ones (29.6 vs 37 µs).
template <int EB, int GAE_L, bool NTL>
__global__ void __launch_bounds__(512) gae_kernel(const float* __restrict__ value, const float* __restrict__ reward,
                                                   const uint8_t* __restrict__ terminal,
                                                   const float* __restrict__ next_value,
                                                   const uint8_t* __restrict__ next_done, int nt, int k, float gamma,
                                                   float gl, int mode, float* __restrict__ adv,
                                                   float* __restrict__ ret, int nts) {
#pragma clang fp contract(off)
  extern __shared__ double sm[];
  const int S = blockDim.x / EB;
  const int el = threadIdx.x % EB, seg = threadIdx.x / EB;
  const int e = blockIdx.x * EB + el;
  const bool ev = e < nt;
  const int lo = seg * GAE_L;
  double* smD = sm;
  double* smC = sm + S * EB;

  float v[GAE_L + 1], r[GAE_L];
  uint32_t tm[GAE_L];
#pragma unroll
  for (int i = 0; i < GAE_L; ++i) {
    const int t = lo + i;
    const bool ok = ev && t < k;
    const size_t idx = (size_t)e + (size_t)nt * t;
    v[i] = ok ? (NTL ? __builtin_nontemporal_load(value + idx) : value[idx]) : 0.0f;
    r[i] = ok ? (NTL ? __builtin_nontemporal_load(reward + idx) : reward[idx]) : 0.0f;
    // done flag that gates step t is terminal[t+1]; beyond the buffer it is next_done (ppo.jl:176)
    tm[i] = ok ? ((t + 1 < k) ? (NTL ? __builtin_nontemporal_load(terminal + idx + nt) : terminal[idx + nt]) : (next_done ? next_done[e] : (uint8_t)0)) : (uint8_t)0;
  }
  {
    const int t = lo + GAE_L;
    v[GAE_L] = (ev && t < k) ? (NTL ? __builtin_nontemporal_load(value + (size_t)e + (size_t)nt * t) : value[(size_t)e + (size_t)nt * t]) : 0.0f;
  }
  // the bootstrap value follows the LAST buffered step (ppo.jl:174 hcat(value, next_values'))
  const float nv = (ev && next_value) ? next_value[e] : 0.0f;

  // δ_t and c_t of step lo+i from the registers (recomputed in phase 2 rather than kept: 64 fewer VGPRs)
  auto step_terms = [&](int i, double& delta, double& cc) {
    const int t = lo + i;
    const float vnext = (t + 1 < k) ? v[i + 1] : nv;
    const double nonterm = 1.0 - (double)(tm[i] ? 1 : 0);
    delta = (double)r[i] + ((double)gamma * nonterm) * (double)vnext - (double)v[i];
    cc = (double)gl * nonterm;
    // compat (ppo.jl:66): the loop starts at k-1 (1-based) so the last slot keeps carry 0 and is defined as 0
    if (t >= k) { delta = 0.0; cc = 1.0; }                                   // padding steps: identity map
    else if (mode == CRL_GAE_COMPAT && t == k - 1) { delta = 0.0; cc = 0.0; }
  };
  // phase 1: compose the segment's affine map A_lo = D + C·A_in
  double D = 0.0, Cc = 1.0;
#pragma unroll
  for (int i = GAE_L - 1; i >= 0; --i) {
    double delta, c;
    step_terms(i, delta, c);
    D = delta + c * D;
    Cc = c * Cc;
  }
  smD[seg * EB + el] = D;
  smC[seg * EB + el] = Cc;
  __syncthreads();
  // keep the raw f32/u8 registers (not their Float64 conversions) alive across the barrier: the conversions are
  // redone in phase 2, which halves the register footprint
#pragma unroll
  for (int i = 0; i < GAE_L; ++i) asm volatile("" : "+v"(v[i]), "+v"(r[i]), "+v"(tm[i]));
  asm volatile("" : "+v"(v[GAE_L]));
  // fold the segments above this one (later in time) into the incoming carry
  double A = 0.0;
  for (int s = S - 1; s > seg; --s) A = smD[s * EB + el] + smC[s * EB + el] * A;
  // phase 2: replay from the carry, fused returns = advantages + value (ppo.jl:181)
#pragma unroll
  for (int i = GAE_L - 1; i >= 0; --i) {
    const int t = lo + i;
    double delta, c;
    step_terms(i, delta, c);
    A = delta + (c * A);
    if (ev && t < k) {
      const size_t idx = (size_t)e + (size_t)nt * t;
      const float a32 = (float)A;
      if (nts) {
        __builtin_nontemporal_store(a32, adv + idx);
        if (ret) __builtin_nontemporal_store(a32 + v[i], ret + idx);
      } else {
        adv[idx] = a32;
        if (ret) ret[idx] = a32 + v[i];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Streaming flavour for batches past the Infinity Cache (crl_gae on big host arrays, crl_gae_bench: 0.57 / 1.14 GB per launch).
// One thread owns E consecutive envs (E = 4: 16-byte accesses, 2: 8-byte, 1: 4-byte) for the whole rollout and walks it backwards with
// the carry in registers: no segments, no LDS, the reference's serial Float64 recurrence in its own order (ppo.jl:63-69) — bit-identical
// to orc_gae (the segmented kernel composes affine maps: <= 1e-6 of its outputs differ in the last bit).
// ROLLING WINDOW (round 5): the inputs of the next D steps are always in flight — slot i of a D-entry register ring is refilled with
// step t - D right after step t has been computed from it — so loads, arithmetic and stores interleave step by step. Round 4's kernel
// moved chunks of eight steps through two register sets (206 VGPRs, two waves per SIMD) and every wave alternated between a burst of 24
// loads and a burst of 16 stores: 0.63-0.66 of 8 TB/s at 524288 envs and 0.43 at 262144. The rolling form (E = 2, D = 8: 72 VGPRs) runs
// 0.72-0.73 at both sizes next to a copy ceiling of 6.4-6.5 TB/s = 0.81 (scripts/micro/stream_rate.hip, profiles/r05_gae_stream_micro.txt).
// ------------------------------------------------------------------------------------------------------------------------------
typedef float gf4 __attribute__((ext_vector_type(4)));
typedef float gf2 __attribute__((ext_vector_type(2)));
typedef float gf1 __attribute__((ext_vector_type(1)));
template <int E> struct GaeVec;
template <> struct GaeVec<4> { typedef gf4 V; typedef uint32_t Dn; };
template <> struct GaeVec<2> { typedef gf2 V; typedef uint16_t Dn; };
template <> struct GaeVec<1> { typedef gf1 V; typedef uint8_t Dn; };
template <int E, int D, bool NTL>
__global__ void __launch_bounds__(256) gae_stream_kernel(const float* __restrict__ value, const float* __restrict__ reward,
                                                         const uint8_t* __restrict__ terminal, const float* __restrict__ next_value,
                                                         const uint8_t* __restrict__ next_done, int nt, int k, float gamma, float gl, int mode,
                                                         float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
  typedef typename GaeVec<E>::V V; typedef typename GaeVec<E>::Dn Dn;
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
  if (e >= (size_t)nt) return;
  double A[E];
  V vnext;                                      // value of step t + 1 (bootstrap behind the last step, ppo.jl:174)
  Dn dnext = 0;                                 // the done bytes that gate step t: terminal[t + 1] (next_done behind the last step, ppo.jl:176)
#pragma unroll
  for (int j = 0; j < E; ++j) { A[j] = 0.0; vnext[j] = 0.0f; }
  if (next_value) vnext = *reinterpret_cast<const V*>(next_value + e);
  if (next_done) dnext = *reinterpret_cast<const Dn*>(next_done + e);
  V sv[D], sr[D]; Dn st[D];
  auto load = [&](int slot, int t) {
    if (t >= 0) {
      const size_t idx = e + (size_t)nt * t;
      const V* pv = reinterpret_cast<const V*>(value + idx); const V* pr = reinterpret_cast<const V*>(reward + idx);
      const Dn* pt = reinterpret_cast<const Dn*>(terminal + idx);
      sv[slot] = NTL ? __builtin_nontemporal_load(pv) : *pv; sr[slot] = NTL ? __builtin_nontemporal_load(pr) : *pr;
      st[slot] = NTL ? __builtin_nontemporal_load(pt) : *pt;
    }
  };
#pragma unroll
  for (int i = 0; i < D; ++i) load(i, k - 1 - i);            // slot of step t = (k - 1 - t) % D
  for (int tb = k - 1; tb >= 0; tb -= D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int t = tb - i;
      if (t >= 0) {
        V a32, r32;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const double nonterm = 1.0 - (double)((((uint32_t)dnext >> (8 * j)) & 0xFFu) ? 1 : 0);
          double delta = (double)sr[i][j] + ((double)gamma * nonterm) * (double)vnext[j] - (double)sv[i][j];
          double cc = (double)gl * nonterm;
          if (mode == CRL_GAE_COMPAT && t == k - 1) { delta = 0.0; cc = 0.0; }   // ppo.jl:66: the loop starts at k-1; slot k is defined as 0
          A[j] = delta + (cc * A[j]);
          a32[j] = (float)A[j];
          r32[j] = a32[j] + sv[i][j];
        }
        const size_t idx = e + (size_t)nt * t;
        if (NTL) {      // the streaming flavour streams both ways: outputs this large are not re-read from a cache either
          __builtin_nontemporal_store(a32, reinterpret_cast<V*>(adv + idx));
          if (ret) __builtin_nontemporal_store(r32, reinterpret_cast<V*>(ret + idx));
        } else {
          *reinterpret_cast<V*>(adv + idx) = a32;
          if (ret) *reinterpret_cast<V*>(ret + idx) = r32;
        }
        vnext = sv[i]; dnext = st[i];
        load(i, t - D);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The segmented scan with TWO envs per thread (8-byte accesses): gae_kernel's arithmetic element by element — the same affine maps composed in
// the same order, so its outputs are bit-identical to gae_kernel's — but every load and store moves 512 bytes per wave instead of 256. At the size
// BASELINE's metric is quoted on (65536 envs x 128 steps, 143 MB, caches flushed) gae_kernel runs 41 µs whatever its tile shape while the library's
// own copy of the same bytes takes 25: with 4 bytes per lane the memory pipeline, not HBM, is the limit (the streaming kernel has known this since
// round 4, but it needs >= 262144 envs to have enough waves). δ_t and c_t are kept from phase 1 for phase 2 here (L = 8: 64 registers).
// ------------------------------------------------------------------------------------------------------------------------------
template <int EB, int GAE_L, bool NTL>
__global__ void __launch_bounds__(512) gae_seg2_kernel(const float* __restrict__ value, const float* __restrict__ reward,
                                                        const uint8_t* __restrict__ terminal, const float* __restrict__ next_value,
                                                        const uint8_t* __restrict__ next_done, int nt, int k, float gamma, float gl, int mode,
                                                        float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
  extern __shared__ double sm[];
  constexpr int E = 2;
  const int S = blockDim.x / EB;
  const int el = threadIdx.x % EB, seg = threadIdx.x / EB;
  const int e = (blockIdx.x * EB + el) * E;                    // first of this thread's two envs (nt is even: both or neither exist)
  const bool ev = e < nt;
  const int lo = seg * GAE_L;
  double* smD = sm;                                            // [S][EB][E]
  double* smC = sm + S * EB * E;
  gf2 v[GAE_L + 1], r[GAE_L];
  uint32_t tm[GAE_L];
  const gf2 zero2 = {0.0f, 0.0f};
#pragma unroll
  for (int i = 0; i < GAE_L; ++i) {
    const int t = lo + i;
    const bool ok = ev && t < k;
    const size_t idx = (size_t)e + (size_t)nt * t;
    const gf2* pv = reinterpret_cast<const gf2*>(value + idx); const gf2* pr = reinterpret_cast<const gf2*>(reward + idx);
    v[i] = ok ? (NTL ? __builtin_nontemporal_load(pv) : *pv) : zero2;
    r[i] = ok ? (NTL ? __builtin_nontemporal_load(pr) : *pr) : zero2;
    const uint16_t* pt = reinterpret_cast<const uint16_t*>(terminal + idx + nt);
    tm[i] = ok ? ((t + 1 < k) ? (uint32_t)(NTL ? __builtin_nontemporal_load(pt) : *pt)
                              : (next_done ? (uint32_t)*reinterpret_cast<const uint16_t*>(next_done + e) : 0u)) : 0u;
  }
  {
    const int t = lo + GAE_L;
    const gf2* pv = reinterpret_cast<const gf2*>(value + (size_t)e + (size_t)nt * t);
    v[GAE_L] = (ev && t < k) ? (NTL ? __builtin_nontemporal_load(pv) : *pv) : zero2;
  }
  const gf2 nv = (ev && next_value) ? *reinterpret_cast<const gf2*>(next_value + e) : zero2;
  double dl[GAE_L][E], cl[GAE_L][E];
  double D[E] = {0.0, 0.0}, Cc[E] = {1.0, 1.0};
#pragma unroll
  for (int i = GAE_L - 1; i >= 0; --i) {
    const int t = lo + i;
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const float vnext = (t + 1 < k) ? v[i + 1][q] : nv[q];
      const double nonterm = 1.0 - (double)(((tm[i] >> (8 * q)) & 0xFFu) ? 1 : 0);
      double delta = (double)r[i][q] + ((double)gamma * nonterm) * (double)vnext - (double)v[i][q];
      double cc = (double)gl * nonterm;
      if (t >= k) { delta = 0.0; cc = 1.0; }
      else if (mode == CRL_GAE_COMPAT && t == k - 1) { delta = 0.0; cc = 0.0; }
      dl[i][q] = delta; cl[i][q] = cc;
      D[q] = delta + cc * D[q];
      Cc[q] = cc * Cc[q];
    }
  }
#pragma unroll
  for (int q = 0; q < E; ++q) { smD[(seg * EB + el) * E + q] = D[q]; smC[(seg * EB + el) * E + q] = Cc[q]; }
  __syncthreads();
  double A[E] = {0.0, 0.0};
  for (int s = S - 1; s > seg; --s) {
#pragma unroll
    for (int q = 0; q < E; ++q) A[q] = smD[(s * EB + el) * E + q] + smC[(s * EB + el) * E + q] * A[q];
  }
#pragma unroll
  for (int i = GAE_L - 1; i >= 0; --i) {
    const int t = lo + i;
    gf2 a32, r32;
#pragma unroll
    for (int q = 0; q < E; ++q) {
      A[q] = dl[i][q] + (cl[i][q] * A[q]);
      a32[q] = (float)A[q];
      r32[q] = a32[q] + v[i][q];
    }
    if (ev && t < k) {
      const size_t idx = (size_t)e + (size_t)nt * t;
      if (NTL) {
        __builtin_nontemporal_store(a32, reinterpret_cast<gf2*>(adv + idx));
        if (ret) __builtin_nontemporal_store(r32, reinterpret_cast<gf2*>(ret + idx));
      } else {
        *reinterpret_cast<gf2*>(adv + idx) = a32;
        if (ret) *reinterpret_cast<gf2*>(ret + idx) = r32;
      }
    }
  }
}

int launch_gae(hipStream_t st, const float* value, const float* reward, const uint8_t* terminal,
               const float* next_value, const uint8_t* next_done, int nt, int k, float gamma, float lambda, int mode,
               float* adv, float* ret, hipEvent_t ev_start, hipEvent_t ev_stop, int seg, int tile, int nt_loads) {
  if (nt <= 0 || k <= 0) { set_error("gae: empty input"); return 1; }
  const float gl = gamma * lambda;  // Float32 product, as `γ * λ` with both T=Float32 (ppo.jl:68)
  // tile = 4 / 2 / 1 selects the streaming kernel with that many envs per thread (seg = its window depth: 0 -> 8, 4, 8 or 16); tile = 0 takes
  // it by itself (two envs per thread, depth 8) for batches past the Infinity Cache
  auto aligned = [&](int E) {
    const uintptr_t fa = (uintptr_t)(4 * E) - 1, da = (uintptr_t)E - 1;
    return (nt % E) == 0 && ((uintptr_t)value & fa) == 0 && ((uintptr_t)reward & fa) == 0 && ((uintptr_t)adv & fa) == 0 && (!ret || ((uintptr_t)ret & fa) == 0) &&
           ((uintptr_t)terminal & da) == 0 && (!next_value || ((uintptr_t)next_value & fa) == 0) && (!next_done || ((uintptr_t)next_done & da) == 0);
  };
  const bool forced = tile == 4 || tile == 2 || tile == 1;
  if (forced && !aligned(tile)) { set_error("gae: the streaming kernel (gae_tile = 4 / 2 / 1) needs num_envs % gae_tile == 0 and buffers aligned to gae_tile floats"); return 1; }
  // measured (profiles/r05_gae_beyond_cache.txt; launch-to-launch spread about 2 %): at 524288 envs x 128 four envs per thread with a window of 4
  // steps 0.73 of 8 TB/s (two envs / window 8: 0.72; the segmented kernel 0.44-0.50), at 262144 envs two envs per thread / window 4 0.70 (four envs: 0.68;
  // segmented 0.49-0.53); at 65536 envs (143 MB: 256-512 waves of this kernel) the segmented kernel stays ahead (0.44 against 0.24 with caches flushed)
  const size_t samples = (size_t)nt * (size_t)k;
  const int auto_E = samples >= ((size_t)1 << 26) ? 4 : 2;
  const bool automatic = tile == 0 && aligned(auto_E) && samples >= ((size_t)1 << 25) && nt >= 262144;
  if (forced || automatic) {
    const int E = forced ? tile : auto_E, Dw = (seg == 4 || seg == 8 || seg == 16) ? seg : (forced ? 8 : 4);
    const dim3 grid((unsigned)((nt / E + 255) / 256)), block(256);
#define CRL_GAE_STREAM(e_, d_)                                                                                                                                   \
    if (E == e_ && Dw == d_) {                                                                                                                                   \
      if (nt_loads) hipExtLaunchKernelGGL((gae_stream_kernel<e_, d_, true>), grid, block, 0, st, ev_start, ev_stop, 0, value, reward, terminal, next_value, next_done, nt, k, gamma, gl, mode, adv, ret); \
      else hipExtLaunchKernelGGL((gae_stream_kernel<e_, d_, false>), grid, block, 0, st, ev_start, ev_stop, 0, value, reward, terminal, next_value, next_done, nt, k, gamma, gl, mode, adv, ret);      \
    } else
    CRL_GAE_STREAM(4, 4) CRL_GAE_STREAM(4, 8) CRL_GAE_STREAM(4, 16) CRL_GAE_STREAM(2, 4) CRL_GAE_STREAM(2, 8) CRL_GAE_STREAM(2, 16)
    CRL_GAE_STREAM(1, 4) CRL_GAE_STREAM(1, 8) CRL_GAE_STREAM(1, 16)
    { set_error("gae: unsupported streaming configuration"); return 1; }
#undef CRL_GAE_STREAM
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  // tile = 128 / 256: the two-envs-per-thread segmented kernel with 32 / 64 env pairs per block (gae_seg2_kernel); L as below
  // Automatic (tile = 0) from 4096 envs on when the buffers allow it — measured against gae_kernel at 128 steps, µs per launch, caches flushed / warm
  // (profiles/r05_gae_pair_kernel.txt): 4096 envs 8.6 vs 10.8 / 6.1 vs 6.6; 8192: 10.0 vs 14.4 / 6.8 vs 12.0; 16384: 14.1 vs 20.2 / 9.2 vs 13.8; 65536: 47.6 vs 50.7
  // / 26.3 vs 30.0, with nontemporal loads 32-34 vs 41-42 (0.56-0.59 of 8 TB/s instead of 0.43; the copy of the same bytes: 25); 1024 envs: 5.6 vs 4.6 (stays).
  if ((tile == 128 || tile == 256) && !aligned(2)) { set_error("gae: gae_tile = 128 / 256 (two envs per thread) needs an even num_envs and 8-byte aligned buffers"); return 1; }
  // (the automatic route is gated on k <= 128, the only depth it was timed at: at k > 128 the pair kernel takes L = 16 — 64 doubles of δ / c and 17 value pairs
  // per thread under launch_bounds(512) — which is parity-tested at k = 300 but was never measured against gae_kernel)
  if (tile == 128 || tile == 256 || (tile == 0 && nt >= 4096 && k <= 128 && aligned(2))) {
    int L2 = (seg == 8 || seg == 16) ? seg : (k <= 128 ? 8 : 16);
    int S2 = (k + L2 - 1) / L2;
    int EB2 = tile == 256 ? 64 : 32;                            // (automatic: 32 pairs per block)
    while (EB2 > 8 && S2 * EB2 > 512) EB2 >>= 1;
    if (S2 * EB2 > 512 && L2 == 8) { L2 = 16; S2 = (k + L2 - 1) / L2; }
    if (S2 * EB2 > 512) { set_error("gae: num_steps > 1024 is not supported"); return 1; }
    const dim3 block2(S2 * EB2), grid2((nt / 2 + EB2 - 1) / EB2);
    const size_t smem2 = sizeof(double) * 2 * S2 * EB2 * 2;
#define CRL_GAE2_CASE(eb, l)                                                                                                              \
    if (EB2 == eb && L2 == l) {                                                                                                             \
      if (nt_loads) hipExtLaunchKernelGGL((gae_seg2_kernel<eb, l, true>), grid2, block2, smem2, st, ev_start, ev_stop, 0, value, reward, terminal, \
                                          next_value, next_done, nt, k, gamma, gl, mode, adv, ret);                                       \
      else hipExtLaunchKernelGGL((gae_seg2_kernel<eb, l, false>), grid2, block2, smem2, st, ev_start, ev_stop, 0, value, reward, terminal,         \
                                 next_value, next_done, nt, k, gamma, gl, mode, adv, ret);                                                \
    } else
    CRL_GAE2_CASE(64, 8) CRL_GAE2_CASE(32, 8) CRL_GAE2_CASE(16, 8) CRL_GAE2_CASE(8, 8) CRL_GAE2_CASE(64, 16) CRL_GAE2_CASE(32, 16) CRL_GAE2_CASE(16, 16) CRL_GAE2_CASE(8, 16)
    { set_error("gae: unsupported pair-tile configuration"); return 1; }
#undef CRL_GAE2_CASE
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  // segment length L and env tile EB: S = ceil(k/L) segments, block = S*EB <= 512 threads.
  // Short segments + narrow tiles give the most loads in flight; long rollouts fall back to longer segments.
  const int env_L = seg == 4 ? 0 : seg, env_EB = tile, env_nts = 0;   // nontemporal STORES measured in round 5 (65536 / 32768 / 131072 envs, caches flushed): no difference for this kernel   // options gae_seg / gae_tile (0 = automatic)
  // (a 4-envs-per-thread variant with 16-B loads was measured SLOWER: 38.9 vs 32.1 us at nt=65536 — 182 VGPRs leave only
  //  2 waves/SIMD; profiles/r01_g_gae_wide_vs_scalar.txt)
  int L = env_L ? env_L : (k <= 256 ? 8 : 16);
  int S = (k + L - 1) / L;
  int EB = env_EB ? env_EB : 64;
  while (EB > 8 && (S * EB > 512 || (nt + EB - 1) / EB < 1024)) EB >>= 1;
  if (S * EB > 512 && L == 8) { L = 16; S = (k + L - 1) / L; }
  if (S * EB > 512) { set_error("gae: num_steps > 1024 is not supported"); return 1; }
  const dim3 block(S * EB), grid((nt + EB - 1) / EB);
  const size_t smem = sizeof(double) * 2 * S * EB;
#define CRL_GAE_CASE(eb, l)                                                                                          \
  if (EB == eb && L == l) {                                                                                          \
    if (nt_loads) hipExtLaunchKernelGGL((gae_kernel<eb, l, true>), grid, block, smem, st, ev_start, ev_stop, 0, value, reward, terminal, \
                                        next_value, next_done, nt, k, gamma, gl, mode, adv, ret, env_nts);           \
    else hipExtLaunchKernelGGL((gae_kernel<eb, l, false>), grid, block, smem, st, ev_start, ev_stop, 0, value, reward, terminal,   \
                               next_value, next_done, nt, k, gamma, gl, mode, adv, ret, env_nts);                    \
  } else
  CRL_GAE_CASE(64, 8) CRL_GAE_CASE(32, 8) CRL_GAE_CASE(16, 8) CRL_GAE_CASE(8, 8)
  CRL_GAE_CASE(64, 16) CRL_GAE_CASE(32, 16) CRL_GAE_CASE(16, 16) CRL_GAE_CASE(8, 16)
  { set_error("gae: unsupported tile configuration"); return 1; }
#undef CRL_GAE_CASE
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// ---- measurement helpers (crl_gae_bench): synthetic inputs of SURVEY §8d's shape and a hand-written same-byte-count copy -------------
__global__ void __launch_bounds__(256) gae_bench_fill_kernel(float* value, float* reward, uint8_t* terminal, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint32_t x = (uint32_t)i * 0x9E3779B9u + (uint32_t)(i >> 32) * 0x85EBCA6Bu + 0x5EEDu;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    value[i] = ((float)(x >> 8) * 0x1.0p-24f - 0.5f) * 20.0f;        // spread like N(0,1)·10
    reward[i] = ((x & 0xFFu) < 5u) ? 0.0f : 1.0f;                    // P(0) ≈ 0.02
    terminal[i] = (((x >> 8) & 0xFFu) < 5u) ? 1 : 0;                 // Bernoulli(≈0.02)
  }
}
// the ceiling: a plain streaming copy, ONE 16-byte piece per thread, nontemporal both ways. scripts/micro/stream_rate.hip swept the grid size
// (512 ... 16384 blocks), 1 / 2 / 4 / 8 pieces in flight per thread, a grid-stride against a block-contiguous walk and cached against
// nontemporal accesses at 0.14 / 0.57 / 1.14 GB: one piece per thread (6.4-6.5 TB/s at 0.57 and 1.14 GB) and eight pieces of a
// block-contiguous share at 16384 blocks (6.4-6.6) lead; round 4's grid-stride walk with four pieces at <= 8192 blocks ran 5.0-5.3.
__global__ void __launch_bounds__(256) gae_bench_copy_kernel(const f32x4v* __restrict__ src, f32x4v* __restrict__ dst, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace crl

// Standalone GAE scan on synthetic device-resident inputs of any size (the handle-bound crl_compute_gae is tied to a rollout buffer):
// `reps` timed launches of gae_kernel with the given flavour, each followed by a timed launch of a hand-written float4 copy that moves
// the same number of bytes (half read, half written). For footprints beyond the 256 MiB Infinity Cache every launch reads from HBM
// (bench.py roofline_gae.beyond_cache); for smaller ones pass flush_mb > 0 and a fill of that size runs before every timed launch.
extern "C" int32_t crl_gae_bench(int32_t device, int32_t nt, int32_t k, int32_t seg, int32_t tile, int32_t nt_loads, int32_t flush_mb,
                                 int32_t reps, double* gae_ms, double* copy_ms) {
  using namespace crl;
  if (nt <= 0 || k <= 0 || reps <= 0 || !gae_ms || !copy_ms) { set_error("crl_gae_bench: bad arguments"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  const size_t n = (size_t)nt * (size_t)k;
  const size_t bytes = 17 * n + 5 * (size_t)nt;
  float *value = nullptr, *reward = nullptr, *adv = nullptr, *ret = nullptr, *nv = nullptr, *flush = nullptr;
  uint8_t *terminal = nullptr, *nd = nullptr;
  f32x4v *csrc = nullptr, *cdst = nullptr;
  const size_t n4 = bytes / 32;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  auto body = [&]() -> int {
    CRL_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CRL_HIP_CHECK(hipEventCreate(&e0)); CRL_HIP_CHECK(hipEventCreate(&e1));
    CRL_HIP_CHECK(hipMalloc(&value, n * 4)); CRL_HIP_CHECK(hipMalloc(&reward, n * 4)); CRL_HIP_CHECK(hipMalloc(&terminal, n));
    CRL_HIP_CHECK(hipMalloc(&adv, n * 4)); CRL_HIP_CHECK(hipMalloc(&ret, n * 4));
    CRL_HIP_CHECK(hipMalloc(&nv, (size_t)nt * 4)); CRL_HIP_CHECK(hipMalloc(&nd, (size_t)nt));
    CRL_HIP_CHECK(hipMalloc(&csrc, n4 * 16)); CRL_HIP_CHECK(hipMalloc(&cdst, n4 * 16));
    CRL_HIP_CHECK(hipMemsetAsync(nv, 0, (size_t)nt * 4, st)); CRL_HIP_CHECK(hipMemsetAsync(nd, 0, (size_t)nt, st));
    CRL_HIP_CHECK(hipMemsetAsync(csrc, 0x3C, n4 * 16, st));
    if (flush_mb > 0) CRL_HIP_CHECK(hipMalloc(&flush, (size_t)flush_mb << 20));
    hipLaunchKernelGGL(gae_bench_fill_kernel, dim3(4096), dim3(256), 0, st, value, reward, terminal, n);
    CRL_HIP_CHECK(hipGetLastError());
    CRL_HIP_CHECK(hipStreamSynchronize(st));
    for (int r = -2; r < reps; ++r) {       // two untimed rounds first
      float ms = 0.f;
      if (flush) CRL_HIP_CHECK(hipMemsetAsync(flush, r & 0xFF, (size_t)flush_mb << 20, st));
      if (launch_gae(st, value, reward, terminal, nv, nd, nt, k, 0.99f, 0.95f, CRL_GAE_FIXED, adv, ret, e0, e1, seg, tile, nt_loads)) return 1;
      CRL_HIP_CHECK(hipEventSynchronize(e1));
      CRL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) gae_ms[r] = ms;
      if (flush) CRL_HIP_CHECK(hipMemsetAsync(flush, (r + 1) & 0xFF, (size_t)flush_mb << 20, st));
      const size_t blocks = (n4 + 255) / 256;
      hipExtLaunchKernelGGL(gae_bench_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, e0, e1, 0, csrc, cdst, n4);
      CRL_HIP_CHECK(hipGetLastError());
      CRL_HIP_CHECK(hipEventSynchronize(e1));
      CRL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) copy_ms[r] = ms;
    }
    return 0;
  };
  rc = body();
  if (st) (void)hipStreamSynchronize(st);
  for (void* p : {(void*)value, (void*)reward, (void*)terminal, (void*)adv, (void*)ret, (void*)nv, (void*)nd, (void*)csrc, (void*)cdst, (void*)flush})
    if (p) (void)hipFree(p);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
