// records.hip — the minibatch gather of ppo.jl:203-211 without six random gathers inside the update kernel.
//   pack:  once per iteration, after GAE: the six per-sample fields the loss closure reads (state, action, logprob, advantage,
//          value, return) → one 64-byte SampleRec per sample, buffer order b = e + nt·t (ppo.jl:184-189). The update kernels
//          fetch recs[b_inds[pos]] themselves: ONE 64-byte line per sample and role (six separate arrays drag six sectors).
//   sums:  Σadv, Σadv² per minibatch of every epoch's permutation (the statistics of ppo.jl:221), Float64, fixed order.
// (Rounds 1-2 also had a permute pass that laid every epoch's minibatches out contiguously — 1.16 ms per iteration and 2.1 GB of
// copies against +2 % on the update kernel for fetching through the permutation; it lost every measurement and is gone.)
#include "common.hpp"
#include "ppo_ctx.hpp"

namespace crl {

#ifndef CRL_PACK_LDS
#define CRL_PACK_LDS 1
#endif
// Round 5: the same records with every global READ 16 bytes per lane. The kernel below gives a sample to four lanes, and the lanes with q = 1 / 2 read the
// five scalar fields — sixteen consecutive floats per wave and instruction, 64-byte transactions; here a block of 256 threads takes 1,024 consecutive samples:
// five waves' worth of 16-byte loads bring the five scalar arrays of the tile into LDS (thread t: floats 4t … 4t + 3 of one array), then every lane assembles its
// quarter as before from LDS and stores it (1 KiB of consecutive records per wave and store instruction, unchanged).
__global__ void __launch_bounds__(256) pack_records_lds_kernel(int B, const float* __restrict__ obs, const int32_t* __restrict__ action,
                                                               const float* __restrict__ logprob, const float* __restrict__ adv,
                                                               const float* __restrict__ value, const float* __restrict__ ret,
                                                               SampleRec* __restrict__ recs) {
  __shared__ __attribute__((aligned(16))) float fld[5][1024];
  __shared__ f32x4 ob[1024];                                    // the tile's observations, brought in by all 256 lanes (sixteen of a wave's lanes used to read them)
  const int q = threadIdx.x & 3;
  const float* arr[5] = {reinterpret_cast<const float*>(action), logprob, adv, value, ret};
  for (int b0 = blockIdx.x * 1024; b0 < B; b0 += gridDim.x * 1024) {
    const bool full = b0 + 1024 <= B && (B & 3) == 0;
    __syncthreads();                                            // the previous tile's readers are done
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int i = 4 * threadIdx.x;
      if (full) *reinterpret_cast<f32x4*>(&fld[a][i]) = *reinterpret_cast<const f32x4*>(arr[a] + b0 + i);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) fld[a][i + e] = (b0 + i + e < B) ? arr[a][b0 + i + e] : 0.0f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int l = 256 * r + threadIdx.x; if (b0 + l < B) ob[l] = reinterpret_cast<const f32x4*>(obs)[b0 + l]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = 64 * r + (threadIdx.x >> 2), b = b0 + l;    // sample of this lane in round r
      if (b < B) {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (q == 0) v = ob[l];
        else if (q == 1) { v[0] = fld[0][l]; v[1] = fld[1][l]; v[2] = fld[2][l]; }
        else if (q == 2) { v[0] = fld[3][l]; v[1] = fld[4][l]; }
        reinterpret_cast<f32x4*>(recs)[(size_t)b * 4 + q] = v;
      }
    }
  }
}

// four lanes per sample, lane q writes quarter q: every store instruction of a wave covers 1 KiB of consecutive records
__global__ void __launch_bounds__(256) pack_records_kernel(int B, const float* __restrict__ obs, const int32_t* __restrict__ action,
                                                           const float* __restrict__ logprob, const float* __restrict__ adv,
                                                           const float* __restrict__ value, const float* __restrict__ ret,
                                                           SampleRec* __restrict__ recs) {
  const int q = threadIdx.x & 3;
  for (int b = blockIdx.x * 64 + (threadIdx.x >> 2); b < B; b += gridDim.x * 64) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (q == 0) v = reinterpret_cast<const f32x4*>(obs)[b];
    else if (q == 1) { v[0] = __int_as_float(action[b]); v[1] = logprob[b]; v[2] = adv[b]; }
    else if (q == 2) { v[0] = value[b]; v[1] = ret[b]; }
    reinterpret_cast<f32x4*>(recs)[(size_t)b * 4 + q] = v;
  }
}

// The advantage statistics of ppo.jl:221 by a gather through the permutation: Σadv, Σadv² per minibatch of every slot; grid (blocks
// per minibatch, num_minibatches, slots), block (bx, mb, z) leaves its sums in part[((z·nmb + mb)·gridDim.x + bx)·2 ..]. adv is the
// 4-byte-per-sample array the GAE wrote (33.5 MB at the headline size: the random reads hit the Infinity Cache); eight independent
// gathers per thread are in flight together. The fallback of the sequential pass (shuffle.hip: adv_bucket_sums_kernel), which needs
// the blocked shuffle's bucket tables.
constexpr int ADVG_U = 8;
__global__ void __launch_bounds__(256) adv_gather_sums_kernel(int B, int M, int chunk, const int32_t* __restrict__ perm /* [slots][B] */,
                                                             const float* __restrict__ adv, double* __restrict__ part) {
  const int mb = blockIdx.y, z = blockIdx.z;
  const int32_t* pm = perm + (size_t)z * B + (size_t)mb * M;
  const int lo = blockIdx.x * chunk, hi = min(M, lo + chunk);
  double sa = 0.0, sa2 = 0.0;
  for (int p0 = lo + threadIdx.x; p0 < hi; p0 += 256 * ADVG_U) {
    int s[ADVG_U];
    float v[ADVG_U];
#pragma unroll
    for (int u = 0; u < ADVG_U; ++u) { const int p = p0 + 256 * u; s[u] = p < hi ? pm[p] : -1; }
#pragma unroll
    for (int u = 0; u < ADVG_U; ++u) v[u] = s[u] >= 0 ? adv[s[u]] : 0.0f;
#pragma unroll
    for (int u = 0; u < ADVG_U; ++u) { const double a = (double)v[u]; sa += a; sa2 += a * a; }
  }
  __shared__ double sm[2][4];
  sa = wave_sum(sa); sa2 = wave_sum(sa2);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = sa; sm[1][threadIdx.x >> 6] = sa2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = part + (((size_t)z * gridDim.y + mb) * gridDim.x + blockIdx.x) * 2;
    o[0] = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
    o[1] = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
  }
}

int launch_adv_fold(crl_ppo* h, const double* part, int nblk, int nentries, double* sums);  // optim.hip

int launch_pack_records(crl_ppo* h) {
  if (!h->recs) { set_error("internal: sample records are used by the fused 4/2/64 path only"); return 1; }
  if (!h->recs_dirty) return 0;
  const int B = h->dc.B;
  int blocks = (B + 63) / 64;
  if (blocks > 8192) blocks = 8192;
  ProfScope ps(h, CRL_K_PACK);
  if (CRL_PACK_LDS) {
    int tiles = (B + 1023) / 1024; if (tiles > 4096) tiles = 4096;
    hipLaunchKernelGGL(pack_records_lds_kernel, dim3(tiles), dim3(256), 0, h->stream, B, h->obs, h->action, h->logprob, h->adv, h->value, h->ret, h->recs);
  } else
  hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(256), 0, h->stream, B, h->obs, h->action, h->logprob, h->adv, h->value, h->ret, h->recs);
  CRL_HIP_CHECK(hipGetLastError());
  h->recs_dirty = false;
  h->slot_fresh = 0;
  return 0;
}

// Σadv, Σadv² of every minibatch of slots [slot0, slot0 + nslots) → adv_sums_base (packs the records first if a field changed)
int launch_slot_adv_sums(crl_ppo* h, int slot0, int nslots) {
  if (launch_pack_records(h)) return 1;
  const int B = h->dc.B, M = h->dc.M, nmb = h->dc.nmb, pb = h->adv_pb;
  const int chunk = (((M + pb - 1) / pb + 63) / 64) * 64;
  double* part = h->adv_part + (size_t)slot0 * nmb * pb * 2;   // each slot has its own slice
  ProfScope ps(h, CRL_K_ADV_STATS);
  if (!opt(h, OPT_ADV_SEQ) || launch_adv_bucket_sums(h, slot0, nslots, part, pb)) {   // the sequential pass needs the blocked shuffle's tables
    hipLaunchKernelGGL(adv_gather_sums_kernel, dim3(pb, nmb, nslots), dim3(256), 0, h->stream, B, M, chunk, h->perm_base + (size_t)slot0 * B,
                       h->adv, part);
  }
  CRL_HIP_CHECK(hipGetLastError());
  if (launch_adv_fold(h, part, pb, nslots * nmb, h->adv_sums_base + (size_t)slot0 * nmb * 2)) return 1;
  for (int s = slot0; s < slot0 + nslots; ++s) h->slot_fresh |= 1u << s;
  return 0;
}

// The speculation guard's snapshot (api.cpp: guard_copy) — thirteen small device buffers saved or restored at the start of every guard
// window — as ONE launch: as thirteen hipMemcpyAsync calls they took ≈ 0.6 ms (blit launches 7-40 µs apart), once per window of 8
// iterations in crl_ppo_iterate and once per iteration in a loop that reads the statistics back after every update.
struct GuardCopyArgs { const void* src[16]; void* dst[16]; unsigned long long bytes[16]; int n; };
__global__ void __launch_bounds__(256) guard_copy_kernel(GuardCopyArgs a) {
  const int p = blockIdx.y;
  if (p >= a.n) return;
  const unsigned long long nb = a.bytes[p];
  const unsigned char* s = static_cast<const unsigned char*>(a.src[p]);
  unsigned char* d = static_cast<unsigned char*>(a.dst[p]);
  const bool al = (((size_t)s | (size_t)d) & 15) == 0;
  const unsigned long long n16 = al ? nb / 16 : 0;
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull)
    reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  for (unsigned long long i = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; i < nb; i += gridDim.x * 256ull) d[i] = s[i];
}
int launch_guard_copy(crl_ppo* h, const void* const* src, void* const* dst, const size_t* bytes, int n) {
  if (n > 16) { set_error("internal: guard snapshot has more than 16 regions"); return 1; }
  GuardCopyArgs a;
  size_t mx = 0;
  for (int i = 0; i < n; ++i) { a.src[i] = src[i]; a.dst[i] = dst[i]; a.bytes[i] = bytes[i]; if (bytes[i] > mx) mx = bytes[i]; }
  a.n = n;
  int bx = (int)((mx / 16 + 255) / 256); if (bx < 1) bx = 1; if (bx > 64) bx = 64;
  hipLaunchKernelGGL(guard_copy_kernel, dim3(bx, n), dim3(256), 0, h->stream, a);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace crl
