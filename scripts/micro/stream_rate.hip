// HBM streaming probe behind the GAE scan's roofline (bench.py roofline_gae.beyond_cache): what does a plain copy reach on this chip,
// and which shape of the streaming GAE kernel (csrc/gae.hip gae_stream_kernel) gets closest to it?
//   copy  : float4 copy of `bytes` (half read, half written) by grid size, loads in flight per thread, nontemporal or not,
//           grid-stride or block-contiguous walk
//   gae   : the serial Float64 recurrence of ppo.jl:63-69, E envs per thread (4: 16-byte accesses, 2: 8-byte), chunks of CH steps, two
//           register sets, nontemporal loads / stores or not — 17 B per (env, step)
// usage: stream_rate [nt=524288] [k=128]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int U, bool NT, bool BLOCKED>
__global__ void __launch_bounds__(256) copy_k(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
  if (BLOCKED) {      // every block owns one contiguous share; inside it the threads stride by 256
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = (lo + per < n4) ? lo + per : n4;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
      for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
    }
    for (; i < hi; i += 256) dst[i] = src[i];
  } else {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
  }
}

template <int E> struct Vec;
template <> struct Vec<4> { typedef f4 V; typedef uint32_t D; };
template <> struct Vec<2> { typedef f2 V; typedef uint16_t D; };
template <int E, int CH> struct Chunk { typename Vec<E>::V v[CH], r[CH]; typename Vec<E>::D t[CH]; };

template <int E, int CH, bool NTL>
__device__ __forceinline__ void chunk_load(Chunk<E, CH>& c, const float* value, const float* reward, const uint8_t* terminal, size_t e, int nt, int t0, int k) {
  typedef typename Vec<E>::V V; typedef typename Vec<E>::D D;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int t = t0 + i;
    if (t < k) {
      const size_t idx = e + (size_t)nt * t;
      const V* pv = reinterpret_cast<const V*>(value + idx); const V* pr = reinterpret_cast<const V*>(reward + idx);
      const D* pt = reinterpret_cast<const D*>(terminal + idx);
      c.v[i] = NTL ? __builtin_nontemporal_load(pv) : *pv; c.r[i] = NTL ? __builtin_nontemporal_load(pr) : *pr; c.t[i] = NTL ? __builtin_nontemporal_load(pt) : *pt;
    }
  }
}

template <int E, int CH, bool NTL, bool NTS, int WPS>
__global__ void __launch_bounds__(256, WPS) gae_k(const float* __restrict__ value, const float* __restrict__ reward, const uint8_t* __restrict__ terminal,
                                                  int nt, int k, float gamma, float gl, float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
  typedef typename Vec<E>::V V; typedef typename Vec<E>::D D;
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
  if (e >= (size_t)nt) return;
  double A[E];
  V vnext; D dnext = 0;
#pragma unroll
  for (int j = 0; j < E; ++j) { A[j] = 0.0; vnext[j] = 0.0f; }
  auto compute = [&](const Chunk<E, CH>& c, int t0) {
#pragma unroll
    for (int i = CH - 1; i >= 0; --i) {
      const int t = t0 + i;
      if (t < k) {
        V a32, r32;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const double nonterm = 1.0 - (double)(((dnext >> (8 * j)) & 0xFFu) ? 1 : 0);
          const double delta = (double)c.r[i][j] + ((double)gamma * nonterm) * (double)vnext[j] - (double)c.v[i][j];
          const double cc = (double)gl * nonterm;
          A[j] = delta + (cc * A[j]);
          a32[j] = (float)A[j]; r32[j] = a32[j] + c.v[i][j];
        }
        const size_t idx = e + (size_t)nt * t;
        if (NTS) { __builtin_nontemporal_store(a32, reinterpret_cast<V*>(adv + idx)); __builtin_nontemporal_store(r32, reinterpret_cast<V*>(ret + idx)); }
        else { *reinterpret_cast<V*>(adv + idx) = a32; *reinterpret_cast<V*>(ret + idx) = r32; }
        vnext = c.v[i]; dnext = c.t[i];
      }
    }
  };
  Chunk<E, CH> ca, cb;
  int t0 = ((k - 1) / CH) * CH;
  chunk_load<E, CH, NTL>(ca, value, reward, terminal, e, nt, t0, k);
  while (true) {
    if (t0 >= CH) chunk_load<E, CH, NTL>(cb, value, reward, terminal, e, nt, t0 - CH, k);
    compute(ca, t0); t0 -= CH; if (t0 < 0) break;
    if (t0 >= CH) chunk_load<E, CH, NTL>(ca, value, reward, terminal, e, nt, t0 - CH, k);
    compute(cb, t0); t0 -= CH; if (t0 < 0) break;
  }
}

// rolling window: D steps' inputs are always in flight; slot i is refilled with step t - D right after step t has been computed from it,
// so loads, Float64 arithmetic and stores interleave step by step instead of chunk by chunk
template <int E, int D, bool NTL, bool NTS, int WPS>
__global__ void __launch_bounds__(256, WPS) gae_roll_k(const float* __restrict__ value, const float* __restrict__ reward, const uint8_t* __restrict__ terminal,
                                                       int nt, int k, float gamma, float gl, float* __restrict__ adv, float* __restrict__ ret) {
#pragma clang fp contract(off)
  typedef typename Vec<E>::V V; typedef typename Vec<E>::D Dn;
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * E;
  if (e >= (size_t)nt) return;
  double A[E];
  V vnext; Dn dnext = 0;
#pragma unroll
  for (int j = 0; j < E; ++j) { A[j] = 0.0; vnext[j] = 0.0f; }
  V sv[D], sr[D]; Dn st[D];
  auto load = [&](int slot, int t) {
    if (t >= 0) {
      const size_t idx = e + (size_t)nt * t;
      const V* pv = reinterpret_cast<const V*>(value + idx); const V* pr = reinterpret_cast<const V*>(reward + idx);
      const Dn* pt = reinterpret_cast<const Dn*>(terminal + idx);
      sv[slot] = NTL ? __builtin_nontemporal_load(pv) : *pv; sr[slot] = NTL ? __builtin_nontemporal_load(pr) : *pr; st[slot] = NTL ? __builtin_nontemporal_load(pt) : *pt;
    }
  };
  // slot of step t = (k - 1 - t) % D: the first D steps (from the end) fill the window
#pragma unroll
  for (int i = 0; i < D; ++i) load(i, k - 1 - i);
  for (int tb = k - 1; tb >= 0; tb -= D) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const int t = tb - i;
      if (t >= 0) {
        V a32, r32;
#pragma unroll
        for (int j = 0; j < E; ++j) {
          const double nonterm = 1.0 - (double)(((dnext >> (8 * j)) & 0xFFu) ? 1 : 0);
          const double delta = (double)sr[i][j] + ((double)gamma * nonterm) * (double)vnext[j] - (double)sv[i][j];
          const double cc = (double)gl * nonterm;
          A[j] = delta + (cc * A[j]);
          a32[j] = (float)A[j]; r32[j] = a32[j] + sv[i][j];
        }
        const size_t idx = e + (size_t)nt * t;
        if (NTS) { __builtin_nontemporal_store(a32, reinterpret_cast<V*>(adv + idx)); __builtin_nontemporal_store(r32, reinterpret_cast<V*>(ret + idx)); }
        else { *reinterpret_cast<V*>(adv + idx) = a32; *reinterpret_cast<V*>(ret + idx) = r32; }
        vnext = sv[i]; dnext = st[i];
        load(i, t - D);
      }
    }
  }
}

__global__ void fill_k(float* value, float* reward, uint8_t* terminal, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    uint32_t x = (uint32_t)i * 0x9E3779B9u + 0x5EEDu; x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    value[i] = ((float)(x >> 8) * 0x1.0p-24f - 0.5f) * 20.0f; reward[i] = ((x & 0xFFu) < 5u) ? 0.0f : 1.0f; terminal[i] = (((x >> 8) & 0xFFu) < 5u) ? 1 : 0;
  }
}

static hipEvent_t e0, e1;
template <typename F> static float timeit(F&& launch, int reps = 7) {
  launch(); launch(); (void)hipDeviceSynchronize();
  float t[16];
  for (int r = 0; r < reps; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&t[r], e0, e1); }
  for (int i = 0; i < reps; ++i) for (int j = i + 1; j < reps; ++j) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  return t[reps / 2];
}

template <int U, bool NT, bool BLOCKED> static void run_copy(const f4* src, f4* dst, size_t n4, int blocks) {
  const float ms = timeit([&] { hipLaunchKernelGGL((copy_k<U, NT, BLOCKED>), dim3(blocks), dim3(256), 0, 0, src, dst, n4); });
  printf("copy  U=%d nt=%d %-8s %6d blocks: %8.1f us  %.2f TB/s\n", U, (int)NT, BLOCKED ? "blocked" : "strided", blocks, ms * 1e3, 2.0 * n4 * 16 / ms / 1e9);
}
template <int E, int CH, bool NTL, bool NTS, int WPS> static void run_gae(const float* v, const float* r, const uint8_t* t, int nt, int k, float* adv, float* ret) {
  const int blocks = (nt / E + 255) / 256;
  const float ms = timeit([&] { hipLaunchKernelGGL((gae_k<E, CH, NTL, NTS, WPS>), dim3(blocks), dim3(256), 0, 0, v, r, t, nt, k, 0.99f, 0.99f * 0.95f, adv, ret); });
  const double bytes = 17.0 * nt * k;
  printf("gae   E=%d CH=%2d ntl=%d nts=%d wps=%d %6d blocks: %8.1f us  %.2f TB/s  %.3f of 8\n", E, CH, (int)NTL, (int)NTS, WPS, blocks, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8);
}

template <int E, int D, bool NTL, bool NTS, int WPS> static void run_roll(const float* v, const float* r, const uint8_t* t, int nt, int k, float* adv, float* ret) {
  const int blocks = (nt / E + 255) / 256;
  const float ms = timeit([&] { hipLaunchKernelGGL((gae_roll_k<E, D, NTL, NTS, WPS>), dim3(blocks), dim3(256), 0, 0, v, r, t, nt, k, 0.99f, 0.99f * 0.95f, adv, ret); });
  const double bytes = 17.0 * nt * k;
  printf("roll  E=%d D=%2d ntl=%d nts=%d wps=%d %6d blocks: %8.1f us  %.2f TB/s  %.3f of 8\n", E, D, (int)NTL, (int)NTS, WPS, blocks, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8);
}

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 524288, k = argc > 2 ? atoi(argv[2]) : 128;
  const size_t n = (size_t)nt * k;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float *value, *reward, *adv, *ret; uint8_t* term; f4 *src, *dst;
  const size_t bytes = 17 * n, n4 = bytes / 32;
  (void)hipMalloc(&value, n * 4); (void)hipMalloc(&reward, n * 4); (void)hipMalloc(&adv, n * 4); (void)hipMalloc(&ret, n * 4); (void)hipMalloc(&term, n);
  (void)hipMalloc(&src, n4 * 16); (void)hipMalloc(&dst, n4 * 16);
  (void)hipMemset(src, 0x3C, n4 * 16);
  hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, value, reward, term, n); (void)hipDeviceSynchronize();
  printf("# nt = %d, k = %d: %.3f GB per launch (17 B per sample)\n", nt, k, bytes / 1e9);
  for (int blocks : {4096, 16384}) { run_copy<8, true, true>(src, dst, n4, blocks); run_copy<4, true, true>(src, dst, n4, blocks); }
  run_copy<1, true, false>(src, dst, n4, (int)((n4 + 255) / 256)); run_copy<1, false, false>(src, dst, n4, (int)((n4 + 255) / 256));
  run_roll<4, 4, true, true, 1>(value, reward, term, nt, k, adv, ret);  run_roll<4, 4, true, true, 4>(value, reward, term, nt, k, adv, ret);
  run_roll<4, 6, true, true, 1>(value, reward, term, nt, k, adv, ret);  run_roll<4, 8, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_roll<4, 8, true, true, 4>(value, reward, term, nt, k, adv, ret);  run_roll<4, 8, false, false, 1>(value, reward, term, nt, k, adv, ret);
  run_roll<4, 8, true, false, 1>(value, reward, term, nt, k, adv, ret); run_roll<4, 12, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_roll<4, 16, true, true, 1>(value, reward, term, nt, k, adv, ret); run_roll<4, 24, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_roll<4, 32, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_roll<2, 8, true, true, 1>(value, reward, term, nt, k, adv, ret);  run_roll<2, 8, true, true, 4>(value, reward, term, nt, k, adv, ret);
  run_roll<2, 16, true, true, 1>(value, reward, term, nt, k, adv, ret); run_roll<2, 16, true, true, 4>(value, reward, term, nt, k, adv, ret);
  run_roll<2, 32, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_gae<4, 8, false, false, 1>(value, reward, term, nt, k, adv, ret); run_gae<4, 8, true, false, 1>(value, reward, term, nt, k, adv, ret);
  run_gae<4, 8, true, true, 1>(value, reward, term, nt, k, adv, ret);   run_gae<4, 8, false, true, 1>(value, reward, term, nt, k, adv, ret);
  run_gae<4, 4, true, true, 1>(value, reward, term, nt, k, adv, ret);   run_gae<4, 4, true, true, 3>(value, reward, term, nt, k, adv, ret);
  run_gae<4, 4, true, true, 4>(value, reward, term, nt, k, adv, ret);   run_gae<4, 4, false, false, 4>(value, reward, term, nt, k, adv, ret);
  run_gae<4, 2, true, true, 4>(value, reward, term, nt, k, adv, ret);   run_gae<4, 16, true, true, 1>(value, reward, term, nt, k, adv, ret);
  run_gae<2, 8, true, true, 1>(value, reward, term, nt, k, adv, ret);   run_gae<2, 8, true, true, 4>(value, reward, term, nt, k, adv, ret);
  run_gae<2, 16, true, true, 2>(value, reward, term, nt, k, adv, ret);  run_gae<2, 16, true, true, 3>(value, reward, term, nt, k, adv, ret);
  run_gae<2, 8, false, false, 4>(value, reward, term, nt, k, adv, ret); run_gae<2, 4, true, true, 4>(value, reward, term, nt, k, adv, ret);
  return 0;
}
