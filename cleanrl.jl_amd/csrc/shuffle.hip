// shuffle.hip — b_inds = shuffle(b_inds) (ppo.jl:191-194) on device.
//  CRL_SHUFFLE_FISHER_YATES: the exact Fisher–Yates loop of Random.shuffle! (for i = n:-1:2, j = rand(1:i), swap),
//    run by one lane, draws from the Philox stream keyed (seed, epoch): bit-identical to orc_shuffle_fy. It is
//    inherently serial (one dependent global swap per element), so it is the parity / small-batch mode.
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

int launch_shuffle(crl_ppo* h, uint64_t epoch_id) {
  const int n = h->dc.B;
  ProfScope ps(h, CRL_K_SHUFFLE);
  if (h->cfg.shuffle_mode == CRL_SHUFFLE_FISHER_YATES) {
    h->perm_is_bijection = false;
    hipLaunchKernelGGL(fy_serial_kernel, dim3(1), dim3(64), 0, h->stream, h->perm, n, h->cfg.seed, epoch_id);
  } else {
    const int bits = bij_bits(n);
    h->perm_epoch = epoch_id; h->perm_is_bijection = true;
    hipLaunchKernelGGL(bijection_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->perm, n, bits, h->cfg.seed, epoch_id);
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
