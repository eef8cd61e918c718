// optim.hip — advantage statistics (ppo.jl:221) and Optimiser(ClipNorm(0.5), Adam(η)) (ppo.jl:93,250).
#include "bijection.hpp"
#include "common.hpp"
#include "ppo_ctx.hpp"
#include "stats.hpp"

namespace crl {

// Σadv, Σadv² per minibatch of the current permutation, Float64, fixed-order two-level sum (reproducible).
// One launch covers all num_minibatches slices; grid = (blocks_per_mb, nmb).
__global__ void adv_sums_kernel(const float* __restrict__ adv, const int32_t* __restrict__ perm, int M,
                                double* __restrict__ part /* [nmb][gridDim.x][2] */) {
  const int mb = blockIdx.y;
  double s = 0.0, s2 = 0.0;
  for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < M; pos += gridDim.x * blockDim.x) {
    const double a = (double)adv[perm[(size_t)mb * M + pos]];
    s += a; s2 += a * a;
  }
  __shared__ double sm[2][16];
  s = wave_sum(s); s2 = wave_sum(s2);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[0][w] = s; sm[1][w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0, t2 = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { t += sm[0][i]; t2 += sm[1][i]; }
    part[((size_t)mb * gridDim.x + blockIdx.x) * 2 + 0] = t;
    part[((size_t)mb * gridDim.x + blockIdx.x) * 2 + 1] = t2;
  }
}
// Same sums when perm is the keyed bijection: read the advantages IN ORDER (coalesced, 4 B per sample instead of a
// 64-B sector per random gather) and ask the inverse bijection which minibatch sample x fell into.
template <int NMB>
__global__ void adv_sums_inv_kernel(const float* __restrict__ adv, int n, int M, int bits, uint64_t seed, uint64_t epoch,
                                    double* __restrict__ part /* [nmb][gridDim.x][2] */) {
  const BijKey key = bij_key(n, bits, seed, epoch);
  double s[NMB], s2[NMB];
#pragma unroll
  for (int m = 0; m < NMB; ++m) { s[m] = 0.0; s2[m] = 0.0; }
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
    const double a = (double)adv[x];
    const int mb = (int)(bij_inverse(key, (uint32_t)x, (uint32_t)n) / (uint32_t)M);
#pragma unroll
    for (int m = 0; m < NMB; ++m) { const bool hit = mb == m; s[m] += hit ? a : 0.0; s2[m] += hit ? a * a : 0.0; }
  }
  __shared__ double sm[2][NMB][16];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < NMB; ++m) {
    const double t = wave_sum(s[m]), t2 = wave_sum(s2[m]);
    if ((threadIdx.x & 63) == 0) { sm[0][m][w] = t; sm[1][m][w] = t2; }
  }
  __syncthreads();
  if (threadIdx.x < NMB) {
    const int m = threadIdx.x;
    double t = 0.0, t2 = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { t += sm[0][m][i]; t2 += sm[1][m][i]; }
    part[((size_t)m * gridDim.x + blockIdx.x) * 2 + 0] = t;
    part[((size_t)m * gridDim.x + blockIdx.x) * 2 + 1] = t2;
  }
}

// one block per minibatch: thread t sums partials t, t+256, … in order, then a fixed tree over the 256 threads
__global__ void __launch_bounds__(256) adv_sums_fold_kernel(const double* __restrict__ part, int nblk, int nmb, double* __restrict__ sums) {
  const int mb = blockIdx.x, t = threadIdx.x;
  if (mb >= nmb) return;
  __shared__ double sm[2][256];
  double a = 0.0, a2 = 0.0;
  for (int b = t; b < nblk; b += 256) { a += part[((size_t)mb * nblk + b) * 2]; a2 += part[((size_t)mb * nblk + b) * 2 + 1]; }
  sm[0][t] = a; sm[1][t] = a2;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (t < w) { sm[0][t] += sm[0][t + w]; sm[1][t] += sm[1][t + w]; }
    __syncthreads();
  }
  if (t == 0) { sums[2 * mb] = sm[0][0]; sums[2 * mb + 1] = sm[1][0]; }
}
// mean / corrected std (StatsBase mean/std, ppo.jl:221) from the (all-reduced) sums; n = global minibatch size
__global__ void adv_finish_kernel(const double* __restrict__ sums, int nentries, double n, double* __restrict__ ms) {
  for (int mb = threadIdx.x; mb < nentries; mb += blockDim.x) {
    const double mean = sums[2 * mb] / n;
    double var = (sums[2 * mb + 1] - n * mean * mean) / (n - 1.0);
    var = var > 0.0 ? var : 0.0;
    ms[2 * mb] = (double)(float)mean;
    ms[2 * mb + 1] = (double)(float)sqrt(var);
  }
}

int launch_adv_fold(crl_ppo* h, const double* part, int nblk, int nentries, double* sums) {
  hipLaunchKernelGGL(adv_sums_fold_kernel, dim3(nentries), dim3(256), 0, h->stream, part, nblk, nentries, sums);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_adv_stats_sums(crl_ppo* h) {
  if (h->bfy_adv_parts > 0) {
    // the blocked Fisher–Yates leaves already summed their slices (shuffle.hip): only the fixed-order fold is left
    ProfScope ps(h, CRL_K_ADV_STATS);
    hipLaunchKernelGGL(adv_sums_fold_kernel, dim3(h->dc.nmb), dim3(256), 0, h->stream, h->bfy_adv_part, h->bfy_adv_parts, h->dc.nmb, h->adv_sums);
    CRL_HIP_CHECK(hipGetLastError());
    h->bfy_adv_parts = 0;
    return 0;
  }
  const int nblk = 64;
  double* part = h->adv_part;   // sized for num_minibatches x 512 blocks in crl_ppo_create
  ProfScope ps(h, CRL_K_ADV_STATS);
  int nfold = nblk;
  if (h->perm_is_bijection && h->dc.nmb == 4) {
    nfold = 512;
    hipLaunchKernelGGL((adv_sums_inv_kernel<4>), dim3(nfold), dim3(512), 0, h->stream, h->adv, h->dc.B, h->dc.M, bij_bits(h->dc.B),
                       shuffle_seed(h), h->perm_epoch, part);
  } else {
    hipLaunchKernelGGL(adv_sums_kernel, dim3(nblk, h->dc.nmb), dim3(512), 0, h->stream, h->adv, h->perm, h->dc.M, part);
  }
  hipLaunchKernelGGL(adv_sums_fold_kernel, dim3(h->dc.nmb), dim3(256), 0, h->stream, part, nfold, h->dc.nmb, h->adv_sums);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}
// mean/std of the minibatches of slots [slot0, slot0 + nslots); slot0 < 0: of the current slot
int launch_adv_stats_finish(crl_ppo* h, int slot0, int nslots) {
  if (slot0 < 0) { slot0 = h->cur_slot; nslots = 1; }
  const size_t o = (size_t)slot0 * h->dc.nmb * 2;
  hipLaunchKernelGGL(adv_finish_kernel, dim3(1), dim3(256), 0, h->stream, h->adv_sums_base + o, nslots * h->dc.nmb,
                     (double)h->dc.M * h->world, h->adv_ms_base + o);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Flux 0.13.4 Optimiser(ClipNorm(thresh), Adam(η, (0.9, 0.999), 1e-8)) applied to each of the 12 parameter arrays
// independently (Q9): one block per array, Float64 scalar math, Float32 state — oracle: orc_clipnorm_adam.
// ------------------------------------------------------------------------------------------------------
struct OptimArgs {
  int off[13]; int arr0;
  float* params; const float* grads; float* m; float* v; double* betap;
  double eta, thresh;
  // data-parallel step: the "Training Statistics" record of the all-reduced message rides in this launch as one extra block
  // (it depends on nothing the optimiser writes) instead of being a launch of its own between the all-reduce and the optimiser
  int stats_block = -1; int P = 0; StatsArgs st{};
};

__global__ void __launch_bounds__(1024) clipnorm_adam_kernel(OptimArgs a) {
#pragma clang fp contract(off)
  if ((int)blockIdx.x == a.stats_block) {
    if (threadIdx.x == 0) compute_stats(a.grads, a.P, a.st.c, a.st.Mglobal, a.st.adv_ms, a.st.mb, a.st.vfix, a.st.out, 0);
    return;
  }
  const int arr = blockIdx.x + a.arr0;
  const int lo = a.off[arr], hi = a.off[arr + 1];
  __shared__ double sm[16];
  double ss = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) { const double g = a.grads[i]; ss += g * g; }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  ss = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) ss += sm[w];
  const float nrm = (float)sqrt(ss);
  const bool clip = (double)nrm > a.thresh;
  const double sc = clip ? a.thresh / (double)nrm : 1.0;
  const double b1 = 0.9, b2 = 0.999, epsn = 1e-8;
  const double bp0 = a.betap[2 * arr], bp1 = a.betap[2 * arr + 1];
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    double g = (double)a.grads[i];
    if (clip) g = (double)(float)(g * sc);
    const float mi = (float)(b1 * (double)a.m[i] + (1 - b1) * g);
    const float vi = (float)(b2 * (double)a.v[i] + (1 - b2) * g * g);
    a.m[i] = mi; a.v[i] = vi;
    const double delta = (double)mi / (1 - bp0) / (sqrt((double)vi / (1 - bp1)) + epsn) * a.eta;
    a.params[i] = a.params[i] - (float)delta;
  }
  __syncthreads();
  if (threadIdx.x == 0) { a.betap[2 * arr] = bp0 * b1; a.betap[2 * arr + 1] = bp1 * b2; }
}

// The same optimiser for large parameter arrays (2×256: W2 has 65,536 entries), where one block per array makes the step a
// 100 µs serial walk: slices of 4,096 entries per block. Two launches: per-slice Σg² partials (Float64; the array's first slice also sets the step's β
// powers aside) → every slice sums its array's partials in slice order (one norm, identical in all slices), clips and applies Adam to its entries with
// the powers set aside, and the array's first slice advances the running powers (a launch of its own until round 5). Same arithmetic per element as
// clipnorm_adam_kernel; only the order of the Σg² sum differs (by slices).
constexpr int OPT_SLICE = 4096;
struct OptimSliceArgs {
  int off[13]; int first_blk[13];   // first_blk[a] = index of array a's first slice; first_blk[12] = number of slices
  float* params; const float* grads; float* m; float* v; double* betap; double* part;
  double eta, thresh;
};
__device__ __forceinline__ void optim_locate(const OptimSliceArgs& a, int blk, int& arr, int& lo, int& hi) {
  arr = 0;
  while (arr < 11 && blk >= a.first_blk[arr + 1]) ++arr;
  const int sl = blk - a.first_blk[arr];
  lo = a.off[arr] + sl * OPT_SLICE;
  hi = lo + OPT_SLICE < a.off[arr + 1] ? lo + OPT_SLICE : a.off[arr + 1];
}
__global__ void __launch_bounds__(1024) clipnorm_partial_kernel(OptimSliceArgs a) {
#pragma clang fp contract(off)
  int arr, lo, hi;
  optim_locate(a, blockIdx.x, arr, lo, hi);
  __shared__ double sm[16];
  double ss = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) { const double g = a.grads[i]; ss += g * g; }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    a.part[blockIdx.x] = t;
    // the β powers this step uses, set aside by the array's first block: adam_slice_kernel reads the copy and its first block of the array advances the
    // original — no block can read a power another one has already advanced (betap_advance_kernel was a launch of its own for that reason)
    if ((int)blockIdx.x == a.first_blk[arr]) { a.part[a.first_blk[12] + 2 * arr] = a.betap[2 * arr]; a.part[a.first_blk[12] + 2 * arr + 1] = a.betap[2 * arr + 1]; }
  }
}
__global__ void __launch_bounds__(1024) adam_slice_kernel(OptimSliceArgs a) {
#pragma clang fp contract(off)
  int arr, lo, hi;
  optim_locate(a, blockIdx.x, arr, lo, hi);
  double ss = 0.0;
  for (int b = a.first_blk[arr]; b < a.first_blk[arr + 1]; ++b) ss += a.part[b];
  const float nrm = (float)sqrt(ss);
  const bool clip = (double)nrm > a.thresh;
  const double sc = clip ? a.thresh / (double)nrm : 1.0;
  const double b1 = 0.9, b2 = 0.999, epsn = 1e-8;
  const double bp0 = a.part[a.first_blk[12] + 2 * arr], bp1 = a.part[a.first_blk[12] + 2 * arr + 1];
  if ((int)blockIdx.x == a.first_blk[arr] && threadIdx.x == 0) { a.betap[2 * arr] = bp0 * b1; a.betap[2 * arr + 1] = bp1 * b2; }
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    double g = (double)a.grads[i];
    if (clip) g = (double)(float)(g * sc);
    const float mi = (float)(b1 * (double)a.m[i] + (1 - b1) * g);
    const float vi = (float)(b2 * (double)a.v[i] + (1 - b2) * g * g);
    a.m[i] = mi; a.v[i] = vi;
    const double delta = (double)mi / (1 - bp0) / (sqrt((double)vi / (1 - bp1)) + epsn) * a.eta;
    a.params[i] = a.params[i] - (float)delta;
  }
}
__global__ void betap_advance_kernel(double* betap) {
  const int arr = threadIdx.x;
  if (arr < 12) { betap[2 * arr] *= 0.9; betap[2 * arr + 1] *= 0.999; }
}

int launch_optim(crl_ppo* h, double eta) {
  OptimArgs a;
  const int hN = h->cfg.hidden, d = h->cfg.obs_dim, A = h->cfg.n_act;
  const int sizes[12] = {hN * d, hN, hN * hN, hN, A * hN, A, hN * d, hN, hN * hN, hN, hN, 1};
  a.off[0] = 0;
  for (int i = 0; i < 12; ++i) a.off[i + 1] = a.off[i] + sizes[i];
  a.params = h->params; a.grads = h->comm_buf; a.m = h->adam_m; a.v = h->adam_v; a.betap = h->betap;
  a.eta = eta; a.thresh = 0.5; a.arr0 = 0;
  ProfScope ps(h, CRL_K_OPTIM);
  if (h->P > 32768 && h->optim_part) {
    OptimSliceArgs b;
    int nb = 0;
    for (int i = 0; i < 12; ++i) { b.off[i] = a.off[i]; b.first_blk[i] = nb; nb += (sizes[i] + OPT_SLICE - 1) / OPT_SLICE; }
    b.off[12] = a.off[12]; b.first_blk[12] = nb;
    b.params = h->params; b.grads = h->comm_buf; b.m = h->adam_m; b.v = h->adam_v; b.betap = h->betap; b.part = h->optim_part;
    b.eta = eta; b.thresh = 0.5;
    hipLaunchKernelGGL(clipnorm_partial_kernel, dim3(nb), dim3(1024), 0, h->stream, b);
    hipLaunchKernelGGL(adam_slice_kernel, dim3(nb), dim3(1024), 0, h->stream, b);
  } else {
    int blocks = 12;
    if (h->stats_pending) {
      a.stats_block = 12; a.P = (int)h->P; blocks = 13;
      a.st.c = h->dc; a.st.Mglobal = (double)h->dc.M * h->world; a.st.adv_ms = h->adv_ms; a.st.mb = h->stats_mb; a.st.vfix = h->vfix;
      a.st.out = h->stats_slot; a.st.fused = 0; a.st.dscale = nullptr;
      h->stats_pending = false;
    }
    hipLaunchKernelGGL(clipnorm_adam_kernel, dim3(blocks), dim3(1024), 0, h->stream, a);
  }
  CRL_HIP_CHECK(hipGetLastError());
  wide_mark_params_changed(h);
  return 0;
}


// Optimiser(ClipNorm(0.5), Adam(η)) over parameter arrays [a0, a1) of a flat 12-array layout (A2C updates the critic's and
// the actor's arrays in separate calls, a2c.jl:88,98)
int launch_clipnorm_adam_range(hipStream_t st, float* params, const float* grads, float* m, float* v, double* betap,
                               const int* off13, int a0, int a1, double eta) {
  OptimArgs a;
  for (int i = 0; i < 13; ++i) a.off[i] = off13[i];
  a.params = params; a.grads = grads; a.m = m; a.v = v; a.betap = betap; a.eta = eta; a.thresh = 0.5; a.arr0 = a0;
  hipLaunchKernelGGL(clipnorm_adam_kernel, dim3(a1 - a0), dim3(1024), 0, st, a);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace crl
