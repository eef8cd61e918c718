// probe.hip — measurement helper, no reference counterpart: the shader clock this GPU sustains under vector load, so that a bench line carries
// its own calibration (boxes of one pool differ by up to 16 % at identical kernels; verdict r5 item 5). Every wave runs independent v_fma_f32
// chains for a fixed span of the constant 100 MHz counter (s_memrealtime) and reports how many shader cycles (s_memtime) passed meanwhile.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ppo_ctx.hpp"

namespace crl {
__global__ void __launch_bounds__(512) clock_probe_kernel(long long span_ticks, double* __restrict__ mhz, float* __restrict__ sink) {
  float a0 = threadIdx.x * 1e-3f, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
  const float m = 0.999f, c = 1e-3f;
  const long long w0 = wall_clock64();
  const unsigned long long c0 = __builtin_readcyclecounter();
  long long w1 = w0;
  do {
#pragma unroll 1
    for (int i = 0; i < 256; ++i) {
      a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
      a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
    }
    w1 = wall_clock64();
  } while (w1 - w0 < span_ticks);
  const unsigned long long c1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) mhz[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (double)(c1 - c0) / ((double)(w1 - w0) / 100.0);
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) sink[0] = a0;   // keeps the chains alive
}
}  // namespace crl

extern "C" int32_t crl_clock_probe(int32_t device, double span_ms, double* median_mhz, double* min_mhz, double* max_mhz) {
  using namespace crl;
  if (!median_mhz || span_ms <= 0.0 || span_ms > 200.0) { set_error("crl_clock_probe: span_ms must be in (0, 200]"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  CRL_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int blocks = cus, waves = blocks * 8;      // one 8-wave block per CU: two waves per SIMD, the update kernel's occupancy
  double* d = nullptr; float* sink = nullptr;
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * waves));
  if (hipMalloc(reinterpret_cast<void**>(&sink), 16) != hipSuccess) { (void)hipFree(d); set_error("crl_clock_probe: hipMalloc failed"); return 1; }
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(512), 0, 0, (long long)(span_ms * 1e5), d, sink);
  std::vector<double> v(waves);
  hipError_t e = hipMemcpy(v.data(), d, sizeof(double) * waves, hipMemcpyDeviceToHost);
  (void)hipFree(d); (void)hipFree(sink);
  if (e != hipSuccess) { set_error(std::string("crl_clock_probe: ") + hipGetErrorString(e)); return 1; }
  std::sort(v.begin(), v.end());
  *median_mhz = v[v.size() / 2];
  if (min_mhz) *min_mhz = v.front();
  if (max_mhz) *max_mhz = v.back();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// crl_product_probe — measurement entry (scripts/product_error.py → profiles/<tag>_product_error.json): C = A·B for the SAME float32
// operands through every way this library (or an f32 BLAS) could multiply them, so that "fp16x2 is narrower than Float32" can be ruled on
// with data (verdict r5 weak 3): the error of each flavour against a Float64 product, on the headline's real operands.
//   flavour 0  fp16x2: split2 (mlp_x2.hpp, the production split) of A·sa and B·sb — or B·col_scale[n], the per-sample scale of the backward
//              product — three v_mfma_f32_32x32x16_f16 per k-step (lo·hi, hi·lo, hi·hi), unscaled in f32
//   flavour 1  bf16x3: split3 (mlp_x3.hpp), six v_mfma_f32_32x32x16_bf16 per k-step
//   flavour 2  v_mfma_f32_32x32x2_f32 (what an f32 GEMM on this GPU's matrix pipe runs)
//   flavour 3  one thread per element, a sequential chain of v_fma_f32 in k order (what Julia's generic matmul / a scalar BLAS kernel does)
// A is [rows][K] row-major, B is [cols][K] (a column's K values contiguous), C comes back as [chunks][cols][rows]: K is cut into `chunks`
// equal pieces (multiples of 16), one f32 accumulator each — the K = samples product of the weight gradient is accumulated per wave and
// folded later, in f32, in the production kernel too; the caller folds the partials.
// ------------------------------------------------------------------------------------------------------
#include "mlp_x2.hpp"

namespace crl {
template <int FLAVOUR>
__global__ void __launch_bounds__(64) product_probe_kernel(const float* __restrict__ A, const float* __restrict__ B, int rows, int cols, int K, int kchunk,
                                                           float sa, float sb, const float* __restrict__ col_scale, float* __restrict__ C) {
  const int lane = threadIdx.x, i = lane & 31, hf = lane >> 5;
  const int rt = blockIdx.x, ct = blockIdx.y, ch = blockIdx.z;
  const int row = 32 * rt + i, col = 32 * ct + i;
  const int k0 = ch * kchunk;
  const float sbc = col_scale ? col_scale[col] : sb;
  if constexpr (FLAVOUR == 3) {
    // 1024 elements of the tile over 64 lanes, 16 each: lane's elements = (rowmap(r, hf), i) like a C fragment
    for (int r = 0; r < 16; ++r) {
      const int rr = 32 * rt + rowmap(r, hf);
      float acc = 0.0f;
      for (int k = 0; k < kchunk; ++k) acc = __builtin_fmaf(A[(size_t)rr * K + k0 + k], B[(size_t)col * K + k0 + k], acc);
      C[((size_t)ch * cols + col) * rows + rr] = acc;
    }
    return;
  }
  f32x16 acc = {};
  if constexpr (FLAVOUR == 2) {
    for (int k = 0; k < kchunk; k += 2) acc = mfma32(A[(size_t)row * K + k0 + k + hf], B[(size_t)col * K + k0 + k + hf], acc);
  } else {
    for (int k = 0; k < kchunk; k += 16) {
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = A[(size_t)row * K + k0 + k + 8 * hf + j];
        b[j] = B[(size_t)col * K + k0 + k + 8 * hf + j];
        if constexpr (FLAVOUR == 0) { a[j] *= sa; b[j] *= sbc; }
      }
      if constexpr (FLAVOUR == 0) acc = mfma_x2(split2(a), split2(b), acc);
      else acc = mfma_x3(split3(a), split3(b), acc);
    }
  }
  float un = 1.0f;
  if constexpr (FLAVOUR == 0) un = 1.0f / (sa * sbc);   // powers of two: exact
  for (int r = 0; r < 16; ++r) C[((size_t)ch * cols + col) * rows + 32 * rt + rowmap(r, hf)] = acc[r] * un;
}
}  // namespace crl

extern "C" int32_t crl_product_probe(int32_t device, int32_t flavour, const float* A, const float* B, int32_t rows, int32_t cols, int32_t K, int32_t chunks,
                                     float scale_a, float scale_b, const float* col_scale, float* C) {
  using namespace crl;
  if (!A || !B || !C || rows <= 0 || cols <= 0 || K <= 0 || chunks <= 0 || rows % 32 || cols % 32 || K % chunks || (K / chunks) % 16 || flavour < 0 || flavour > 3) {
    set_error("crl_product_probe: rows / cols multiples of 32, K / chunks a multiple of 16, flavour 0 … 3"); return 1;
  }
  CRL_HIP_CHECK(hipSetDevice(device));
  const size_t na = (size_t)rows * K, nb = (size_t)cols * K, nc = (size_t)chunks * rows * cols;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr, *dS = nullptr;
  int rc = 0;
  auto fail = [&](hipError_t e, const char* what) { set_error(std::string("crl_product_probe: ") + what + ": " + hipGetErrorString(e)); rc = 1; };
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&dA), na * 4)) != hipSuccess) fail(e, "hipMalloc");
  if (!rc && (e = hipMalloc(reinterpret_cast<void**>(&dB), nb * 4)) != hipSuccess) fail(e, "hipMalloc");
  if (!rc && (e = hipMalloc(reinterpret_cast<void**>(&dC), nc * 4)) != hipSuccess) fail(e, "hipMalloc");
  if (!rc && col_scale && (e = hipMalloc(reinterpret_cast<void**>(&dS), (size_t)cols * 4)) != hipSuccess) fail(e, "hipMalloc");
  if (!rc && (e = hipMemcpy(dA, A, na * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy A");
  if (!rc && (e = hipMemcpy(dB, B, nb * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy B");
  if (!rc && col_scale && (e = hipMemcpy(dS, col_scale, (size_t)cols * 4, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy scales");
  if (!rc) {
    const dim3 grid(rows / 32, cols / 32, chunks), block(64);
    const int kc = K / chunks;
    switch (flavour) {
      case 0: hipLaunchKernelGGL(product_probe_kernel<0>, grid, block, 0, 0, dA, dB, rows, cols, K, kc, scale_a, scale_b, dS, dC); break;
      case 1: hipLaunchKernelGGL(product_probe_kernel<1>, grid, block, 0, 0, dA, dB, rows, cols, K, kc, scale_a, scale_b, dS, dC); break;
      case 2: hipLaunchKernelGGL(product_probe_kernel<2>, grid, block, 0, 0, dA, dB, rows, cols, K, kc, scale_a, scale_b, dS, dC); break;
      default: hipLaunchKernelGGL(product_probe_kernel<3>, grid, block, 0, 0, dA, dB, rows, cols, K, kc, scale_a, scale_b, dS, dC); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) fail(e, "launch");
    if (!rc && (e = hipMemcpy(C, dC, nc * 4, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy C");
  }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dS);
  return rc;
}
