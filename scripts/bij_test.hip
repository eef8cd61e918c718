#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../cleanrl.jl_amd/csrc/bijection.hpp"
using namespace crl;
__global__ void k(int n, int bits, uint32_t* fwd, uint32_t* inv, uint32_t* keyout) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  BijKey key = bij_key(n, bits, 0x5EED, 11);
  if (p == 0) { for (int i = 0; i < 6; ++i) keyout[i] = key.k[i]; for (int i = 0; i < 3; ++i) keyout[6 + i] = key.inv[i]; keyout[9] = key.mask; keyout[10] = key.bits; }
  if (p >= n) return;
  fwd[p] = bij_forward(key, p, n);
  inv[p] = bij_inverse(key, p, n);
}
int main() {
  for (int n : {1024, 2368, 524288}) {
    int bits = bij_bits(n);
    uint32_t *f, *iv, *ko;
    hipMalloc(&f, n * 4); hipMalloc(&iv, n * 4); hipMalloc(&ko, 64);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, bits, f, iv, ko);
    std::vector<uint32_t> hf(n), hi(n); uint32_t hk[16];
    hipMemcpy(hf.data(), f, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), iv, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hk, ko, 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int p = 0; p < n; ++p) if (hi[hf[p]] != (uint32_t)p) ++bad;
    printf("n %d bits %d bad %d  key %08x %08x inv0 %08x check %08x mask %x\n", n, bits, bad, hk[0], hk[1], hk[6], (hk[1] | 1u) * hk[6], hk[9]);
  }
}
