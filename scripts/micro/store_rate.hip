// Store-bandwidth probe: how fast can the chip WRITE? 1 GiB per launch, grid-stride.
//   fill4      float4 per lane, 1 KB contiguous per wave-instruction (plain / nontemporal)
//   fill1      one dword per lane, 256 B contiguous per wave-instruction
//   rows8      float4 per lane, a wave-instruction covers 8 separate 128-B segments 1 KB apart (the tile_out pattern of csrc/wide.hip)
//   scatter16  float4 per lane, 64 lanes in 64 different 1 KB rows (the h1 store pattern of the fused forward's staging)
//   copy4      read + write float4 (half the bytes each way)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(f4* dst, const f4* src, size_t n4) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  const f4 v = {1.0f, 2.0f, 3.0f, (float)tid};
  if (MODE == 0) for (size_t i = tid; i < n4; i += nth) dst[i] = v;
  if (MODE == 1) for (size_t i = tid; i < n4; i += nth) __builtin_nontemporal_store(v, dst + i);
  if (MODE == 2) { float* d = reinterpret_cast<float*>(dst); for (size_t i = tid; i < 4 * n4; i += nth) d[i] = v[0]; }
  if (MODE == 3) {   // wave w covers rows (8 consecutive 1 KB rows), lane = (row = lane >> 3, 16-B piece = lane & 7) of a 128-B column block c
    const size_t wave = tid >> 6; const int lane = tid & 63;
    const size_t nw = nth >> 6, ngroups = n4 / 512;       // a group = 8 rows x 1 KB = 512 float4
    for (size_t g = wave; g < ngroups; g += nw)
      for (int c = 0; c < 8; ++c) dst[g * 512 + (size_t)(lane >> 3) * 64 + c * 8 + (lane & 7)] = v;
  }
  if (MODE == 4) {   // wave covers 64 rows of 1 KB; per instruction each lane writes 16 B of its own row
    const size_t wave = tid >> 6; const int lane = tid & 63;
    const size_t nw = nth >> 6, ngroups = n4 / 4096;      // a group = 64 rows x 1 KB
    for (size_t g = wave; g < ngroups; g += nw)
      for (int c = 0; c < 64; ++c) dst[g * 4096 + (size_t)lane * 64 + c] = v;
  }
  if (MODE == 5) for (size_t i = tid; i < n4 / 2; i += nth) dst[i] = src[n4 / 2 + i];
}
template <int MODE>
static void run(const char* name, f4* buf, size_t n4, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, buf, n4); (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, buf, n4); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double bytes = (double)n4 * 16;
  printf("%-10s %5d blocks: %.3f ms -> %.2f TB/s (%.1f GB/s per CU)\n", name, blocks, best, bytes / best / 1e9, bytes / best / 1e6 / 256);
}
int main() {
  const size_t n4 = (size_t)1 << 26;   // 1 GiB
  f4* buf; (void)hipMalloc(&buf, n4 * 16);
  for (int blocks : {2048, 8192}) {
    run<0>("fill4", buf, n4, blocks); run<1>("fill4-nt", buf, n4, blocks); run<2>("fill1", buf, n4, blocks); run<3>("rows8", buf, n4, blocks);
    run<4>("scatter16", buf, n4, blocks); run<5>("copy4", buf, n4, blocks);
  }
  return 0;
}
