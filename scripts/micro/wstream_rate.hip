// Weight-slab streaming probe (the loop of csrc/wide_fused.hpp without the arithmetic): every block of 512 threads pulls the SAME 256 KB
// (8 slabs of 32 KB) from L2 into LDS over and over, one barrier per slab. Prints µs per slab and GB/s per CU for:
//   dma        global_load_lds_dwordx4, 4 pieces of 1 KB per wave and slab, vmcnt(0) + barrier per slab
//   dma-rot    the same with the piece order rotated per block
//   dma-ahead  two slabs in flight (three buffers, counted vmcnt)
//   reg        global_load_dwordx4 into registers, ds_write_b128, barrier (what wide_dense_x2_kernel does)
//   dma-4w / dma-4w-s / reg-4w   only waves 4-7 fetch (8 pieces each: the producers of wide_fused_fwd_pc_kernel) — compiler-chosen address form,
//              the SGPR-base + VGPR-offset form by inline asm, and through registers; the ISSUE time of the 8 instructions is printed too
//   … each with 1 block per CU (104 KB of LDS) and, where it fits, 2 blocks per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int SLAB = 32768;
template <int MODE>
__global__ void __launch_bounds__(512) k(const float* __restrict__ W, float* out, int tiles, int pad_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smx[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rot = MODE == 1 ? (int)((blockIdx.x * 5u + (blockIdx.x >> 3)) & 31u) : 0;
  float acc = 0.0f;
  auto dma = [&](int slab, unsigned char* dst) {
    const char* g = reinterpret_cast<const char*>(W) + (size_t)slab * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = (i * 8 + wave + rot) & 31;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  if (MODE <= 1) {
    for (int t = 0; t < tiles; ++t)
      for (int s = 0; s < 8; ++s) {
        dma(s, smx + (s & 1) * SLAB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<float*>(smx + (s & 1) * SLAB)[tid];
      }
  } else if (MODE == 2) {
    dma(0, smx); dma(1, smx + SLAB);
    for (int t = 0; t < tiles; ++t)
      for (int s = 0; s < 8; ++s) {
        const int g = t * 8 + s;
        dma((s + 2) & 7, smx + ((g + 2) % 3) * SLAB);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // slab g has landed; g + 1 and g + 2 stay in flight
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<float*>(smx + (g % 3) * SLAB)[tid];
        __builtin_amdgcn_s_barrier();                        // (the read above must finish before buffer g % 3 is refilled two slabs on)
      }
  } else if (MODE >= 4) {
    unsigned long long ticks = 0;
    for (int t = 0; t < tiles; ++t)
      for (int s = 0; s < 8; ++s) {
        unsigned char* dst = smx + (s & 1) * SLAB;
        if (wave >= 4) {
          const int p = wave - 4;
          const char* g = reinterpret_cast<const char*>(W) + (size_t)s * SLAB;
          const unsigned long long t0 = __builtin_readcyclecounter();
          if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int piece = i * 4 + p;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024 + lane * 16),
                                               (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
            }
          } else if (MODE == 5) {
            const unsigned voff = lane * 16;
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int piece = i * 4 + p;
              const char* gb = g + piece * 1024;
              asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(gb), "s"(lds0 + piece * 1024) : "memory");
            }
          } else {
            f4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = reinterpret_cast<const f4*>(g + (i * 4 + p) * 1024)[lane];
            const unsigned long long t1 = __builtin_readcyclecounter();
            ticks += t1 - t0;
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<f4*>(dst + (i * 4 + p) * 1024)[lane] = r[i];
          }
          if (MODE != 6) { const unsigned long long t1 = __builtin_readcyclecounter(); ticks += t1 - t0; }
          asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<float*>(dst)[tid];
      }
    if (wave >= 4 && lane == 0 && blockIdx.x < 8) reinterpret_cast<unsigned long long*>(out + 1024 * 512)[blockIdx.x * 4 + wave - 4] = ticks;
  } else {
    for (int t = 0; t < tiles; ++t)
      for (int s = 0; s < 8; ++s) {
        const f4* g = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(W) + (size_t)s * SLAB) + tid;
        f4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = g[512 * i];
        f4* d = reinterpret_cast<f4*>(smx + (s & 1) * SLAB) + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[512 * i] = r[i];
        __syncthreads();
        acc += reinterpret_cast<float*>(smx + (s & 1) * SLAB)[tid];
      }
  }
  out[(size_t)blockIdx.x * 512 + tid] = acc + (float)pad_lds;
}
template <int MODE>
static void run(const char* name, const float* W, float* out, int blocks, size_t lds) {
  const int tiles = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, W, out, 2, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), lds, 0, W, out, tiles, 0); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_block_slabs = (double)tiles * 8 * (blocks / 256.0);      // slabs a CU processed
  const double us = ms * 1e3 / per_block_slabs;
  printf("%-10s %4d blocks (%d per CU, %3zu KB LDS each): %.3f ms, %.3f us per slab per CU -> %.1f GB/s per CU, %.2f TB/s chip\n", name, blocks, blocks / 256, lds >> 10, ms, us,
         SLAB / us / 1e3, SLAB / us / 1e3 * 256 / 1e3);
  if (MODE >= 4) {
    unsigned long long tk[32]; hipMemcpy(tk, out + 1024 * 512, sizeof(tk), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 32; ++i) sum += (double)tk[i];
    printf("           issue of the 8 fetch instructions: %.0f shader cycles per slab (s_memtime, mean of 32 producer waves)\n", sum / 32 / (tiles * 8.0));
  }
}
int main() {
  float *W, *out; hipMalloc(&W, 8 * SLAB); hipMalloc(&out, 1024 * 512 * 4 + 4096); hipMemset(W, 0x3c, 8 * SLAB);
  run<0>("dma", W, out, 256, 104 << 10); run<1>("dma-rot", W, out, 256, 104 << 10); run<2>("dma-ahead", W, out, 256, 104 << 10); run<3>("reg", W, out, 256, 104 << 10);
  run<0>("dma", W, out, 512, 70 << 10); run<1>("dma-rot", W, out, 512, 70 << 10); run<3>("reg", W, out, 512, 70 << 10);
  run<0>("dma", W, out, 1024, 64 << 10 >> 1); run<3>("reg", W, out, 1024, 64 << 10 >> 1);
  run<4>("dma-4w", W, out, 256, 104 << 10); run<5>("dma-4w-s", W, out, 256, 104 << 10); run<6>("reg-4w", W, out, 256, 104 << 10);
  return 0;
}
