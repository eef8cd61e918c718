// VALU issue-rate probe: v_fma_f32 vs v_pk_fma_f32 vs v_rcp_f32, 1/2/4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x[8]; f32x2 y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i][0] = x[i]; y[i][1] = x[i] + 0.5f; }
  f32x2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
      else if (MODE == 1) y[i] = __builtin_elementwise_fma(y[i], av, bv);
      else x[i] = __builtin_amdgcn_rcpf(x[i]);
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i][0] + y[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps : {1, 2, 4}) {
    for (int mode = 0; mode < 3; ++mode) {
      dim3 grid(256), block(256 * wps);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr_per_wave = (double)iters * 8;
      double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
      printf("waves/SIMD %d mode %s: %.3f ms, %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", wps,
             mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32" : "v_rcp_f32", ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    }
  }
}
