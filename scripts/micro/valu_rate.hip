// VALU issue-rate probe: W waves per SIMD, each running 8 independent v_fma_f32 chains. Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) k(float* out, int iters, long long* cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0000001f, c = 1e-9f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int wps : {1, 2, 3, 4}) {           // waves per SIMD: block = 4 SIMDs x wps waves, one block per CU
    const int threads = 64 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, cyc); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)iters * 64 * wps;
    printf("waves/SIMD %d: %.3f ms, shader clock cycles %lld -> %.2f cycles per wave-instruction per SIMD (%.2f per wave)\n", wps, ms, c,
           (double)c / instr_per_simd, (double)c / ((double)iters * 64));
  }
  return 0;
}
