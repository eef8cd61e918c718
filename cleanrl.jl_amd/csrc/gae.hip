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
// One thread owns FOUR consecutive envs for the whole rollout and walks it backwards in chunks of 8 steps, the carry in registers:
// no segments, no LDS, and every access is 16 bytes per lane (value, reward, advantage, return: 1 KB rows per wave-instruction) or 4
// (the four done bytes of a step) — 40 memory instructions per 544 bytes instead of 41 per 136. The next chunk's 24 loads are issued
// before the current chunk is computed (two register sets). The arithmetic is the reference's serial Float64 recurrence in its own
// order (ppo.jl:63-69), so the result is bit-identical to orc_gae (the segmented kernel composes affine maps: ≤ 1e-6 of its outputs
// differ in the last bit). Needs nt % 4 == 0; pays only when there are enough envs to fill the chip (one thread per four envs): taken by itself from 2^26 samples.
// ------------------------------------------------------------------------------------------------------------------------------
typedef float gf4 __attribute__((ext_vector_type(4)));
struct GaeChunk { gf4 v[8], r[8]; uint32_t t[8]; };
template <bool NTL>
__device__ __forceinline__ void gae_chunk_load(GaeChunk& c, const float* value, const float* reward, const uint8_t* terminal, size_t e, int nt, int t0, int k) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = t0 + i;
    if (t < k) {
      const size_t idx = e + (size_t)nt * t;
      const gf4* pv = reinterpret_cast<const gf4*>(value + idx);
      const gf4* pr = reinterpret_cast<const gf4*>(reward + idx);
      const uint32_t* pt = reinterpret_cast<const uint32_t*>(terminal + idx);
      c.v[i] = NTL ? __builtin_nontemporal_load(pv) : *pv;
      c.r[i] = NTL ? __builtin_nontemporal_load(pr) : *pr;
      c.t[i] = NTL ? __builtin_nontemporal_load(pt) : *pt;
    }
  }
}
template <bool NTL>
__global__ void __launch_bounds__(256) gae_stream_kernel(const float* __restrict__ value, const float* __restrict__ reward,
                                                         const uint8_t* __restrict__ terminal, const float* __restrict__ next_value,
                                                         const uint8_t* __restrict__ next_done, int nt, int k, float gamma, float gl, int mode,
                                                         float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= (size_t)nt) return;
  double A[4] = {0.0, 0.0, 0.0, 0.0};
  gf4 vnext = {0.0f, 0.0f, 0.0f, 0.0f};        // value of step t + 1 (bootstrap behind the last step, ppo.jl:174)
  uint32_t dnext = 0;                           // the done bytes that gate step t: terminal[t + 1] (next_done behind the last step, ppo.jl:176)
  if (next_value) vnext = *reinterpret_cast<const gf4*>(next_value + e);
  if (next_done) dnext = *reinterpret_cast<const uint32_t*>(next_done + e);
  auto compute = [&](const GaeChunk& c, int t0) {
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      const int t = t0 + i;
      if (t < k) {
        gf4 a32, r32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double nonterm = 1.0 - (double)(((dnext >> (8 * j)) & 0xFFu) ? 1 : 0);
          double delta = (double)c.r[i][j] + ((double)gamma * nonterm) * (double)vnext[j] - (double)c.v[i][j];
          double cc = (double)gl * nonterm;
          if (mode == CRL_GAE_COMPAT && t == k - 1) { delta = 0.0; cc = 0.0; }   // ppo.jl:66: the loop starts at k-1; slot k is defined as 0
          A[j] = delta + (cc * A[j]);
          a32[j] = (float)A[j];
          r32[j] = a32[j] + c.v[i][j];
        }
        const size_t idx = e + (size_t)nt * t;
        *reinterpret_cast<gf4*>(adv + idx) = a32;
        if (ret) *reinterpret_cast<gf4*>(ret + idx) = r32;
        vnext = c.v[i]; dnext = c.t[i];
      }
    }
  };
  GaeChunk ca, cb;
  int t0 = ((k - 1) / 8) * 8;
  gae_chunk_load<NTL>(ca, value, reward, terminal, e, nt, t0, k);
  while (true) {
    if (t0 >= 8) gae_chunk_load<NTL>(cb, value, reward, terminal, e, nt, t0 - 8, k);
    compute(ca, t0);
    t0 -= 8;
    if (t0 < 0) break;
    if (t0 >= 8) gae_chunk_load<NTL>(ca, value, reward, terminal, e, nt, t0 - 8, k);
    compute(cb, t0);
    t0 -= 8;
    if (t0 < 0) break;
  }
}

int launch_gae(hipStream_t st, const float* value, const float* reward, const uint8_t* terminal,
               const float* next_value, const uint8_t* next_done, int nt, int k, float gamma, float lambda, int mode,
               float* adv, float* ret, hipEvent_t ev_start, hipEvent_t ev_stop, int seg, int tile, int nt_loads) {
  if (nt <= 0 || k <= 0) { set_error("gae: empty input"); return 1; }
  const float gl = gamma * lambda;  // Float32 product, as `γ * λ` with both T=Float32 (ppo.jl:68)
  // tile = 4 selects the streaming kernel (four envs per thread); tile = 0 takes it by itself for batches past the Infinity Cache
  const bool can_stream = (nt % 4) == 0 && ((uintptr_t)value % 16) == 0 && ((uintptr_t)reward % 16) == 0 && ((uintptr_t)adv % 16) == 0 &&
                          (!ret || ((uintptr_t)ret % 16) == 0) && ((uintptr_t)terminal % 4) == 0 && (!next_value || ((uintptr_t)next_value % 16) == 0) &&
                          (!next_done || ((uintptr_t)next_done % 4) == 0);
  if (tile == 4 && !can_stream) { set_error("gae: the streaming kernel (gae_tile = 4) needs num_envs % 4 == 0 and 16-byte aligned buffers"); return 1; }
  // measured (profiles/r04_gae_beyond_cache.txt): at 524288 envs x 128 (1.14 GB) the streaming kernel runs at the float4 copy's own speed
  // (216-226 us, 0.63-0.66 of 8 TB/s; the segmented kernel 282-342 us); at 262144 envs its 1024 waves are too few (160-176 us against 128-135)
  if (tile == 4 || (tile == 0 && can_stream && (size_t)nt * (size_t)k >= ((size_t)1 << 26) && nt >= 262144)) {
    const dim3 grid((unsigned)((nt / 4 + 255) / 256)), block(256);
    if (nt_loads) hipExtLaunchKernelGGL((gae_stream_kernel<true>), grid, block, 0, st, ev_start, ev_stop, 0, value, reward, terminal, next_value, next_done, nt, k, gamma, gl, mode, adv, ret);
    else hipExtLaunchKernelGGL((gae_stream_kernel<false>), grid, block, 0, st, ev_start, ev_stop, 0, value, reward, terminal, next_value, next_done, nt, k, gamma, gl, mode, adv, ret);
    CRL_HIP_CHECK(hipGetLastError());
    return 0;
  }
  // segment length L and env tile EB: S = ceil(k/L) segments, block = S*EB <= 512 threads.
  // Short segments + narrow tiles give the most loads in flight; long rollouts fall back to longer segments.
  const int env_L = seg, env_EB = tile, env_nts = 0;   // options gae_seg / gae_tile (0 = automatic)
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
// the ceiling: a plain streaming copy, 16 B per lane, four pieces in flight per thread, nontemporal both ways
__global__ void __launch_bounds__(256) gae_bench_copy_kernel(const f32x4v* __restrict__ src, f32x4v* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4v a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
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
      size_t blocks = (n4 + 256 * 4 - 1) / (256 * 4);
      if (blocks > 8192) blocks = 8192;
      if (blocks < 1) blocks = 1;
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
