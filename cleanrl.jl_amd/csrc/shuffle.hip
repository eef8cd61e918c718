// shuffle.hip — b_inds = shuffle(b_inds) (ppo.jl:191-194) on device.
//  CRL_SHUFFLE_FISHER_YATES: the exact Fisher–Yates loop of Random.shuffle! (for i = n:-1:2, j = rand(1:i), swap),
//    run by one lane, draws from the Philox stream keyed (seed, epoch): bit-identical to orc_shuffle_fy. It is
//    inherently serial (one dependent global swap per element), so it is the parity / small-batch mode.
//  CRL_SHUFFLE_BLOCKED_FY: the same distribution (uniform over S_n), parallel: Rao–Sandelius split into ~64-element
//    sub-buckets + Fisher–Yates inside each (see below). Exact and deterministic; ~3x the cost of the bijection.
//  CRL_SHUFFLE_BIJECTION: throughput mode — perm[p] = π_key(p) with π a keyed bijection of [0,n) (invertible
//    multiply / xorshift rounds on ceil(log2 n) bits + cycle walking). O(1) per element, no dependence between
//    elements, one coalesced store each. A pseudo-random permutation, not a uniform draw from S_n.
#include "bijection.hpp"
#include "common.hpp"
#include "ppo_ctx.hpp"

namespace crl {

__global__ void fy_serial_kernel(int32_t* __restrict__ perm, int n, uint64_t seed, uint64_t epoch) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int i = n - 1; i >= 1; --i) {
    const u32x4 o = philox((uint32_t)i, (uint32_t)epoch, (uint32_t)(epoch >> 32), 0x5FFu, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t r = ((uint64_t)o.x << 32) | o.y;
    const uint32_t j = (uint32_t)__umul64hi(r, (uint64_t)(i + 1));
    const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
  }
}

__global__ void iota_kernel(int32_t* __restrict__ perm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = i;
}

__global__ void bijection_kernel(int32_t* __restrict__ perm, int n, int bits, uint64_t seed, uint64_t epoch) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const BijKey key = bij_key(n, bits, seed, epoch);
  perm[p] = (int32_t)bij_forward(key, (uint32_t)p, (uint32_t)n);
}

// ------------------------------------------------------------------------------------------------------
// CRL_SHUFFLE_BLOCKED_FY — an exact, parallel, deterministic shuffle (uniform over S_n like ppo.jl:194's shuffle):
// Rao–Sandelius split + Fisher–Yates leaves. Every element draws two random digits (d1 < K1, d2 < 256) from its own
// Philox counter; elements are scattered to their L1 bucket (global, LDS-aggregated atomics), each L1 bucket (≈4 K
// elements) is split by d2 inside LDS, and every sub-bucket (≈16 elements) is first put in ascending order by a
// parallel rank count — which makes the result independent of the order the atomics happened to serve — and then
// shuffled by the textbook Fisher–Yates loop run by one lane. Sub-buckets are concatenated in (d1,d2) order.
// Bit-identical to orc_shuffle_blocked_fy.
// ------------------------------------------------------------------------------------------------------
constexpr int BFY_L1 = 4096;        // expected elements per L1 bucket
constexpr int BFY_CAP = 5632;       // LDS capacity of one L1 bucket (+24 sigma)
constexpr int BFY_MAXK1 = 16384;    // batches up to 2^26 samples

__device__ __forceinline__ void bfy_digits(uint32_t i, uint32_t K1, uint64_t seed, uint64_t epoch, uint32_t& d1, uint32_t& d2) {
  const u32x4 o = philox(i, (uint32_t)epoch, (uint32_t)(epoch >> 32), 0xB0Cu, (uint32_t)seed, (uint32_t)(seed >> 32));
  d1 = o.x & (K1 - 1u); d2 = o.y & 255u;
}

// Scatter pass: S[d1·CAP + (position inside bucket d1)] = i with LDS-aggregated reservations on the bucket's cursor. Buckets are
// PADDED to their LDS capacity (BFY_CAP = expected 4096 + 24 sigma), so no counting pass has to run first to find where a bucket
// starts: the scan after this pass turns the cursors into output offsets for the leaf pass. (The count pass was 0.18 of the 1.0 ms
// the four epochs' shuffles take; the permutation does not depend on the scatter order — the leaves sort canonically.)
// All kernels of the blocked shuffle take the epoch from blockIdx.y: crl_ppo_iterate draws the permutations of all
// update_epochs in ONE launch per pass (epoch e works in workspace slice e, perm slot e) — 4x fewer, 4x larger launches.
constexpr int BFY_WS_STRIDE = 4 * 16384 + 8;   // u32 words of workspace per epoch: (unused) | off | cur | err

// One block covers 8192 elements (1024 threads x 8, or 256 x 32 for small shards so that the grid still covers the chip): measured on
// one box at n = 8.4 M x 4 epochs, 64 / 32 / 16 / 8 / 4 elements per thread of a 1024-thread block took 0.96 / 0.87 / 0.80 / 0.76 / 0.76 ms
// for the whole shuffle — the pass is bound by latency (LDS atomics, the reservation round trip), which more and smaller blocks hide;
// the 2 K global reservations per block are not what it waits for. ONE LDS atomic per element: its return value is the element's rank
// inside the block's share of the bucket and rides in a register to the store. When n <= 2^24 the element's second digit travels in
// the top byte of its S entry, so the leaf pass needs no Philox call to find it again.
template <int BFY_T1>
__global__ void __launch_bounds__(BFY_T1) bfy_l1_kernel(int n, uint32_t K1, uint64_t seed, uint64_t epoch0, uint32_t* __restrict__ ws,
                                                     int32_t* __restrict__ S0, size_t sstride, int packed, uint16_t* __restrict__ dig1_0) {
  constexpr int EPT = 8192 / BFY_T1;
  const uint64_t epoch = epoch0 + blockIdx.y;
  uint32_t* tot = ws + (size_t)blockIdx.y * BFY_WS_STRIDE;
  uint32_t* cur = tot + 2 * BFY_MAXK1 + 1;
  uint32_t* err = cur + BFY_MAXK1;
  int32_t* S = S0 + (size_t)blockIdx.y * sstride;
  // the element's first digit for adv_bucket_sums_kernel (round 5): that pass recomputed it — one Philox call per sample and epoch, most of its 108 µs on the
  // iteration's critical path — although this kernel has it in a register; 2 bytes per sample and epoch
  uint16_t* dig1 = dig1_0 ? dig1_0 + (size_t)blockIdx.y * n : nullptr;
  extern __shared__ uint32_t lds[];   // hist[K1] → (after the reservations) the block's base inside every bucket
  uint32_t* hist = lds;
  for (uint32_t d = threadIdx.x; d < K1; d += BFY_T1) hist[d] = 0;
  __syncthreads();
  uint32_t dig[EPT];                  // d1 (14 bits) | rank inside the block's share (13 bits) << 14 ; 0xFFFFFFFF = past the end
  uint32_t d2p[EPT / 4];              // the second digits, four to a register
  const int i0 = blockIdx.x * 8192 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < EPT / 4; ++q) d2p[q] = 0;
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int i = i0 + q * BFY_T1;
    uint32_t d1, d2, w = 0xFFFFFFFFu;
    if (i < n) {
      bfy_digits((uint32_t)i, K1, seed, epoch, d1, d2);
      w = d1 | (atomicAdd(&hist[d1], 1u) << 14);
      d2p[q >> 2] |= d2 << (8 * (q & 3));
      if (dig1) dig1[i] = (uint16_t)d1;
    }
    dig[q] = w;
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < K1; d += BFY_T1) { const uint32_t c = hist[d]; hist[d] = c ? atomicAdd(&cur[d], c) : 0u; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < EPT; ++q) {
    const int i = i0 + q * BFY_T1;
    if (i < n) {
      const uint32_t d1 = dig[q] & 0x3FFFu;
      const uint32_t at = hist[d1] + (dig[q] >> 14);
      const uint32_t d2 = (d2p[q >> 2] >> (8 * (q & 3))) & 255u;
      if (at < (uint32_t)BFY_CAP) S[(size_t)d1 * BFY_CAP + at] = (int32_t)((uint32_t)i | (packed ? d2 << 24 : 0u));
      else *err = 1u;                       // the bucket does not fit its leaf (the scan flags it too); never written out of bounds
    }
  }
}

// exclusive scan of the K1 bucket totals (one block); cur = off; flags buckets that do not fit the LDS leaf kernel
__global__ void __launch_bounds__(1024) bfy_scan_kernel(uint32_t K1, uint32_t* __restrict__ ws) {
  uint32_t* off = ws + (size_t)blockIdx.x * BFY_WS_STRIDE + BFY_MAXK1;
  const uint32_t* tot = off + BFY_MAXK1 + 1;                 // the scatter pass's cursors = bucket sizes
  uint32_t* err = const_cast<uint32_t*>(tot) + BFY_MAXK1;
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const int per = (K1 + 1023) / 1024;
  uint32_t s = 0;
  for (int q = 0; q < per; ++q) { const uint32_t d = t * per + q; if (d < K1) { s += tot[d]; if (tot[d] > (uint32_t)BFY_CAP) *err = 1u; } }
  part[t] = s;
  __syncthreads();
  if (t == 0) { uint32_t a = 0; for (int q = 0; q < 1024; ++q) { const uint32_t v = part[q]; part[q] = a; a += v; } }
  __syncthreads();
  uint32_t a = part[t];
  for (int q = 0; q < per; ++q) { const uint32_t d = t * per + q; if (d < K1) { off[d] = a; a += tot[d]; } }
  if (t == 1023) off[K1] = a;
}

// leaves: one block per L1 bucket; 256 sub-buckets ↔ 256 lanes for the Fisher–Yates tail. No per-thread arrays and no
// unrolling: a low register count keeps several blocks per CU resident. The Fisher–Yates draws do not depend on the data, so every
// thread computes the draws of the positions it owns (16 Philox calls each, no divergence) while it ranks them; the serial tail
// that one lane runs per sub-bucket is then swaps only. (The tail used to call Philox per swap: a wave waited for its longest
// sub-bucket, ≈27 dependent Philox chains for 16 useful ones, and the pass took 0.48 ms for four epochs of 8.4 M.)
// With adv != nullptr the block also leaves Σadv, Σadv² of its slice per minibatch in part[mb][bucket][2] (Float64, fixed
// order): the advantage statistics of ppo.jl:221 then need no separate gather pass over the permutation. Requires
// M >= BFY_CAP so that a bucket touches at most two minibatches.
// threads of a leaf block: LDS allows two blocks per CU, so the thread count is what hides latency — 256 / 512 / 1024 threads: 0.72 / 0.58 /
// 0.53 ms for the four epochs' shuffles at n = 8.4 M (one box, nothing else running); 52 registers, no scratch
constexpr int BFY_LT = 1024;
__global__ void __launch_bounds__(BFY_LT, 2) bfy_leaf_kernel(uint32_t K1, uint64_t seed, uint64_t epoch0, int n, const uint32_t* __restrict__ ws,
                                                          const int32_t* __restrict__ S0, size_t sstride, int32_t* __restrict__ perm0,
                                                          const float* __restrict__ adv, int M, int nmb, double* __restrict__ part,
                                                          uint16_t* __restrict__ bucket_mb0, uint8_t* __restrict__ mbid0, int packed) {
  const uint64_t epoch = epoch0 + blockIdx.y;
  const uint32_t* off = ws + (size_t)blockIdx.y * BFY_WS_STRIDE + BFY_MAXK1;
  const uint32_t* err = off + 2 * BFY_MAXK1 + 1;
  const int32_t* S = S0 + (size_t)blockIdx.y * sstride + (size_t)blockIdx.x * BFY_CAP;   // this bucket's padded slice
  int32_t* perm = perm0 + (size_t)blockIdx.y * n;
  if (*err) return;
  __shared__ int32_t buf[BFY_CAP], buf2[BFY_CAP];
  __shared__ uint16_t draw[BFY_CAP];                // draw[p]: the Fisher–Yates partner of position p inside its sub-bucket
  __shared__ uint8_t dig[BFY_CAP], dsub[BFY_CAP];   // sub-bucket of the element at a pre-scatter / post-scatter position
  __shared__ uint32_t cnt[256], soff[257], run[256];
  const uint32_t d1 = blockIdx.x;
  const uint32_t base = off[d1], c = off[d1 + 1] - base;
  const int t = threadIdx.x;
  if (t < 256) { cnt[t] = 0; run[t] = 0; }
  __syncthreads();
#pragma unroll 1
  for (uint32_t idx = t; idx < c; idx += BFY_LT) {
    int32_t v = S[idx];
    uint32_t a, d2;
    if (packed) { d2 = (uint32_t)v >> 24; v &= 0xFFFFFF; }
    else bfy_digits((uint32_t)v, K1, seed, epoch, a, d2);
    buf2[idx] = v; dig[idx] = (uint8_t)d2;
    atomicAdd(&cnt[d2], 1u);
  }
  __syncthreads();
  if (t < 64) {   // exclusive scan of 256 counts by one wave: 4 per lane + a wave prefix
    const uint32_t c0 = cnt[4 * t], c1 = cnt[4 * t + 1], c2 = cnt[4 * t + 2], c3 = cnt[4 * t + 3];
    uint32_t incl = c0 + c1 + c2 + c3;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if (t >= o) incl += y; }
    const uint32_t excl = incl - (c0 + c1 + c2 + c3);
    soff[4 * t] = excl; soff[4 * t + 1] = excl + c0; soff[4 * t + 2] = excl + c0 + c1; soff[4 * t + 3] = excl + c0 + c1 + c2;
    if (t == 63) soff[256] = incl;
  }
  __syncthreads();
#pragma unroll 1
  for (uint32_t idx = t; idx < c; idx += BFY_LT) {
    const uint32_t d2 = dig[idx];
    const uint32_t pos = soff[d2] + atomicAdd(&run[d2], 1u);
    buf[pos] = buf2[idx];
    dsub[pos] = (uint8_t)d2;     // travels with the element: the rank pass below needs no search for the sub-bucket that owns a position
  }
  __syncthreads();
  // canonical order: every element counts the smaller members of its sub-bucket (values are distinct) and moves there; and the
  // draw of the Fisher–Yates step that will visit this POSITION (Random.shuffle!: for i = n:-1:2, swap with rand(1:i))
#pragma unroll 2
  for (uint32_t idx = t; idx < c; idx += BFY_LT) {
    const int32_t v = buf[idx];
    const uint32_t d2 = dsub[idx];          // the sub-bucket that owns position idx: soff[d2] <= idx < soff[d2+1]
    const uint32_t lo = soff[d2], hi = soff[d2 + 1];
    const uint32_t j = idx - lo;
    uint32_t x = 0;
    if (j >= 1) {
      const u32x4 o = philox(d1 * 256u + d2, j, (uint32_t)epoch ^ 0x9E3779B9u, (uint32_t)(epoch >> 32) ^ 0xF15A7E5u, (uint32_t)seed,
                             (uint32_t)(seed >> 32));
      const uint64_t r = ((uint64_t)o.x << 32) | o.y;
      x = (uint32_t)__umul64hi(r, (uint64_t)(j + 1));
    }
    draw[idx] = (uint16_t)x;
    uint32_t r = 0;
    for (uint32_t p = lo; p < hi; ++p) r += (buf[p] < v) ? 1u : 0u;
    buf2[lo + r] = v;
  }
  __syncthreads();
  if (t < 256) {
    int32_t* m = buf2 + soff[t];
    const uint16_t* dr = draw + soff[t];
    const int n2 = (int)cnt[t];
#pragma unroll 1
    for (int j = n2 - 1; j >= 1; --j) {
      const uint32_t x = dr[j];
      const int32_t tmp = m[j]; m[j] = m[x]; m[x] = tmp;
    }
  }
  __syncthreads();
  if (!adv) {
    for (uint32_t idx = t; idx < c; idx += BFY_LT) perm[base + idx] = buf2[idx];
    if (bucket_mb0) {
      // which minibatch this bucket's positions [base, base + c) belong to; a bucket across a boundary also says it per member
      const uint32_t mbf = base / (uint32_t)M, cutp = (mbf + 1u) * (uint32_t)M;
      const bool straddle = c > 0 && base + c > cutp;
      if (t == 0) bucket_mb0[(size_t)blockIdx.y * BFY_MAXK1 + d1] = (uint16_t)(mbf | (straddle ? 0x8000u : 0u));
      if (straddle) {
        uint8_t* mbid = mbid0 + (size_t)blockIdx.y * n;
        for (uint32_t idx = t; idx < c; idx += BFY_LT) mbid[buf2[idx]] = (uint8_t)((base + idx) / (uint32_t)M);
      }
    }
    return;
  }
  const int mb0 = (int)(base / (uint32_t)M);
  const uint32_t cut = (uint32_t)(mb0 + 1) * (uint32_t)M;     // first position of the next minibatch
  // phase 1: all gathers in flight together (results parked in the dead `buf`), phase 2: ordered sums from LDS
  float* abuf = reinterpret_cast<float*>(buf);
#pragma unroll 8
  for (uint32_t idx = t; idx < c; idx += BFY_LT) {
    const int32_t v = buf2[idx];
    perm[base + idx] = v;
    abuf[idx] = adv[v];
  }
  double sa = 0.0, sa2 = 0.0, sb = 0.0, sb2 = 0.0;
  for (uint32_t idx = t; idx < c; idx += BFY_LT) {
    const double a = (double)abuf[idx];
    if (base + idx < cut) { sa += a; sa2 += a * a; } else { sb += a; sb2 += a * a; }
  }
  __shared__ double red[BFY_LT / 64][4];
  sa = wave_sum(sa); sa2 = wave_sum(sa2); sb = wave_sum(sb); sb2 = wave_sum(sb2);
  if ((t & 63) == 0) { red[t >> 6][0] = sa; red[t >> 6][1] = sa2; red[t >> 6][2] = sb; red[t >> 6][3] = sb2; }
  __syncthreads();
  if (t < nmb) {
    double v0 = 0.0, v1 = 0.0;
    const int o = t == mb0 ? 0 : 2;
    if (t == mb0 || t == mb0 + 1)
      for (int w = 0; w < BFY_LT / 64; ++w) { v0 += red[w][o]; v1 += red[w][o + 1]; }     // fixed order
    part[((size_t)t * K1 + d1) * 2] = v0; part[((size_t)t * K1 + d1) * 2 + 1] = v1;
  }
}

// Permutations of epochs [epoch_id, epoch_id + nslots) into perm slots [cur_slot, cur_slot + nslots), one launch per pass.
// fused (single slot only) = also leave the per-minibatch advantage sums in h->bfy_adv_part (layer-wise path: there the
// advantages are final before the shuffle runs)
static int launch_blocked_fy(crl_ppo* h, uint64_t epoch_id, int nslots, bool fused) {
  const int n = h->dc.B;
  uint32_t K1 = 1;
  while ((uint64_t)K1 * (uint64_t)BFY_L1 < (uint64_t)n) K1 *= 2;
  if (K1 > (uint32_t)BFY_MAXK1) { set_error("blocked Fisher-Yates supports batches up to 2^26 samples"); return 1; }
  if (h->cur_slot + nslots > h->cfg.update_epochs) { set_error("internal: permutation slots out of range"); return 1; }
  uint32_t* ws = h->bfy_ws + (size_t)h->cur_slot * BFY_WS_STRIDE;
  const size_t sstride = (size_t)K1 * BFY_CAP;
  int32_t* S = h->perm_tmp + (size_t)h->cur_slot * sstride;
  // the buckets' cursors of all slots in one strided fill
  CRL_HIP_CHECK(hipMemset2DAsync(ws + 2 * BFY_MAXK1 + 1, sizeof(uint32_t) * BFY_WS_STRIDE, 0, sizeof(uint32_t) * K1, (size_t)nslots, h->stream));
  const uint64_t seed = shuffle_seed(h);
  const bool big = n >= (4 << 20);
  const int chunks = (n + 8191) / 8192, packed = n <= (1 << 24) ? 1 : 0;
  const dim3 g1(chunks, nslots);
  const bool fuse = fused && nslots == 1 && h->bfy_adv_part && h->dc.M >= BFY_CAP && h->dc.nmb <= 256;
  // the bucket → minibatch tables (sequential advantage statistics, below) describe slots [0, nslots) of one iterate call
  const bool tables = !fuse && h->bfy_bucket_mb && h->cur_slot == 0 && h->dc.nmb <= 255;
  uint16_t* dig1 = (tables && h->bfy_dig1) ? h->bfy_dig1 : nullptr;
  if (big) hipLaunchKernelGGL((bfy_l1_kernel<1024>), g1, dim3(1024), sizeof(uint32_t) * K1, h->stream, n, K1, seed, epoch_id, ws, S, sstride, packed, dig1);
  else hipLaunchKernelGGL((bfy_l1_kernel<256>), g1, dim3(256), sizeof(uint32_t) * K1, h->stream, n, K1, seed, epoch_id, ws, S, sstride, packed, dig1);
  hipLaunchKernelGGL(bfy_scan_kernel, dim3(nslots), dim3(1024), 0, h->stream, K1, ws);
  h->bfy_dig1_valid = dig1 != nullptr;
  hipLaunchKernelGGL(bfy_leaf_kernel, dim3(K1, nslots), dim3(BFY_LT), 0, h->stream, K1, seed, epoch_id, n, ws, S, sstride, h->perm, fuse ? h->adv : nullptr,
                     h->dc.M, h->dc.nmb, h->bfy_adv_part, tables ? h->bfy_bucket_mb : nullptr, h->bfy_mbid, packed);
  CRL_HIP_CHECK(hipGetLastError());
  h->bfy_adv_parts = fuse ? (int)K1 : 0;
  for (int z = 0; z < nslots; ++z) h->bfy_tbl_slots &= ~(1u << (h->cur_slot + z));
  if (tables) {   // exactly these slots: older tables were drawn for another epoch0
    h->bfy_tbl_slots = nslots >= 32 ? 0xFFFFFFFFu : ((1u << nslots) - 1u); h->bfy_tbl_epoch0 = epoch_id; h->bfy_tbl_K1 = K1;
  }
  return 0;
}

// Σadv, Σadv² per minibatch of the blocked-Fisher–Yates permutations in slots [slot0, slot0 + nslots) WITHOUT walking the
// permutations: a sample's position is decided by its first Philox digit (its L1 bucket) up to the order inside the bucket, and a
// bucket (≈4 K consecutive positions) lies inside one minibatch unless it crosses one of the nmb − 1 boundaries. So one coalesced
// pass over adv recomputes every sample's digit, looks its bucket's minibatch up in the table the leaf pass left, and only for the
// members of the few straddling buckets reads the per-sample byte. 4 B read per sample and epoch instead of a random cache line.
template <int NMB>
__global__ void __launch_bounds__(256) adv_bucket_sums_kernel(int n, uint32_t K1, uint64_t seed, uint64_t epoch0, int slot0,
                                                             const float* __restrict__ adv, const uint16_t* __restrict__ bucket_mb0,
                                                             const uint8_t* __restrict__ mbid0, double* __restrict__ part /* [z][mb][gridDim.x][2] */,
                                                             const uint16_t* __restrict__ dig1_0) {
  const int z = blockIdx.z;
  const uint64_t epoch = epoch0 + (uint64_t)(slot0 + z);
  const uint16_t* bmb = bucket_mb0 + (size_t)(slot0 + z) * BFY_MAXK1;
  const uint8_t* mbid = mbid0 + (size_t)(slot0 + z) * n;
  // the epoch's bucket table in LDS: out of global memory the per-lane lookup (64 different 2-byte addresses inside a 4 KB table per wave and sample row) cost the
  // address unit a cycle per lane — most of this kernel's 108 µs (round 5)
  extern __shared__ uint16_t lbmb[];
  for (uint32_t i = threadIdx.x; i < K1; i += 256) lbmb[i] = bmb[i];
  __syncthreads();
  double s[NMB], s2[NMB];
#pragma unroll
  for (int m = 0; m < NMB; ++m) { s[m] = 0.0; s2[m] = 0.0; }
  if (dig1_0 && (n & 3) == 0) {
    // four consecutive samples per thread: 16 bytes of advantages and 8 bytes of stored digits per lane and load (one sample per thread: 4 + 2 bytes)
    const uint16_t* dg = dig1_0 + (size_t)(slot0 + z) * n;
    typedef float af4 __attribute__((ext_vector_type(4)));
    typedef unsigned short ud4 __attribute__((ext_vector_type(4)));
    for (int v = 4 * (blockIdx.x * 256 + threadIdx.x); v < n; v += gridDim.x * 1024) {
      const af4 a4 = *reinterpret_cast<const af4*>(adv + v);
      const ud4 d4 = *reinterpret_cast<const ud4*>(dg + v);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double a = (double)a4[q];
        const uint32_t e = lbmb[d4[q]];
        const int mb = (e & 0x8000u) ? (int)mbid[v + q] : (int)e;
#pragma unroll
        for (int m = 0; m < NMB; ++m) { const bool hit = mb == m; s[m] += hit ? a : 0.0; s2[m] += hit ? a * a : 0.0; }
      }
    }
  } else
  for (int v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256) {
    const double a = (double)adv[v];
    uint32_t d1, d2;
    if (dig1_0) d1 = dig1_0[(size_t)(slot0 + z) * n + v];       // left by bfy_l1_kernel
    else bfy_digits((uint32_t)v, K1, seed, epoch, d1, d2);
    const uint32_t e = lbmb[d1];
    const int mb = (e & 0x8000u) ? (int)mbid[v] : (int)e;
#pragma unroll
    for (int m = 0; m < NMB; ++m) { const bool hit = mb == m; s[m] += hit ? a : 0.0; s2[m] += hit ? a * a : 0.0; }
  }
  __shared__ double sm[2][NMB][4];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < NMB; ++m) {
    const double t = wave_sum(s[m]), t2 = wave_sum(s2[m]);
    if ((threadIdx.x & 63) == 0) { sm[0][m][w] = t; sm[1][m][w] = t2; }
  }
  __syncthreads();
  if (threadIdx.x < NMB) {
    const int m = threadIdx.x;
    double* o = part + (((size_t)z * NMB + m) * gridDim.x + blockIdx.x) * 2;
    o[0] = (sm[0][m][0] + sm[0][m][1]) + (sm[0][m][2] + sm[0][m][3]);
    o[1] = (sm[1][m][0] + sm[1][m][1]) + (sm[1][m][2] + sm[1][m][3]);
  }
}

// returns 1 when the tables do not cover these slots (other shuffle modes, caller-supplied permutations, nmb not 1/2/4/8): the
// caller then gathers through the permutation instead
int launch_adv_bucket_sums(crl_ppo* h, int slot0, int nslots, double* part, int nblk) {
  if (h->cfg.shuffle_mode != CRL_SHUFFLE_BLOCKED_FY || !h->bfy_bucket_mb) return 1;
  for (int z = slot0; z < slot0 + nslots; ++z) if (!(h->bfy_tbl_slots >> z & 1u)) return 1;
  const int nmb = h->dc.nmb;
  const dim3 g(nblk, 1, nslots);
#define CRL_GO(N) hipLaunchKernelGGL((adv_bucket_sums_kernel<N>), g, dim3(256), sizeof(uint16_t) * h->bfy_tbl_K1, h->stream, h->dc.B, h->bfy_tbl_K1, shuffle_seed(h), \
                                     h->bfy_tbl_epoch0, slot0, h->adv, h->bfy_bucket_mb, h->bfy_mbid, part, (h->bfy_dig1_valid && opt(h, OPT_ADV_SEQ) == 1) ? h->bfy_dig1 : nullptr)
  if (nmb == 1) CRL_GO(1); else if (nmb == 2) CRL_GO(2); else if (nmb == 4) CRL_GO(4); else if (nmb == 8) CRL_GO(8); else return 1;
#undef CRL_GO
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// update_epochs consecutive epochs at once into slots 0 … (crl_ppo_iterate; blocked Fisher–Yates only)
int launch_shuffle_epochs(crl_ppo* h, uint64_t epoch0, int nslots) {
  ProfScope ps(h, CRL_K_SHUFFLE);
  h->bfy_adv_parts = 0; h->perm_is_bijection = false;
  return launch_blocked_fy(h, epoch0, nslots, false);
}

int launch_shuffle(crl_ppo* h, uint64_t epoch_id, bool with_adv_sums) {
  const int n = h->dc.B;
  ProfScope ps(h, CRL_K_SHUFFLE);
  h->bfy_adv_parts = 0;
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_BLOCKED_FY) {
    h->perm_is_bijection = false;
    return launch_blocked_fy(h, epoch_id, 1, with_adv_sums);
  }
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_FISHER_YATES) {
    h->perm_is_bijection = false;
    hipLaunchKernelGGL(fy_serial_kernel, dim3(1), dim3(64), 0, h->stream, h->perm, n, shuffle_seed(h), epoch_id);
  } else {
    const int bits = bij_bits(n);
    h->perm_epoch = epoch_id; h->perm_is_bijection = true;
    hipLaunchKernelGGL(bijection_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->perm, n, bits, shuffle_seed(h), epoch_id);
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_iota(crl_ppo* h) {
  hipLaunchKernelGGL(iota_kernel, dim3((h->dc.B + 255) / 256), dim3(256), 0, h->stream, h->perm, h->dc.B);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace crl
