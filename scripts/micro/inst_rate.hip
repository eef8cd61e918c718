// Issue cost of instruction classes on gfx950: W waves per SIMD, each running 8 independent chains of ONE instruction. Prints ns per
// wave-instruction per SIMD for 1, 2 and 4 waves per SIMD — the table DESIGN.md §3 weighs the update kernel's instruction mix with.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/inst_rate scripts/micro/inst_rate.hip && /tmp/inst_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHAIN8(INS)                                                                                                     \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                  \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))

#define KERNEL32(NAME, INS)                                                                                             \
  __global__ void __launch_bounds__(1024) NAME(float* out, int iters) {                                                 \
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const float b = 1.0000001f, c = 1e-9f;                                                                              \
    for (int i = 0; i < iters; ++i) {                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) CHAIN8(INS);                                                        \
    }                                                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
  }

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_CVTPK(n) "v_cvt_pk_f16_f32 %" #n ", %" #n ", %8\n"
#define I_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
KERNEL32(k_fma, I_FMA)
KERNEL32(k_mullo, I_MULLO)
KERNEL32(k_mulhi, I_MULHI)
KERNEL32(k_exp, I_EXP)
KERNEL32(k_rcp, I_RCP)
KERNEL32(k_log, I_LOG)
KERNEL32(k_xor, I_XOR)
KERNEL32(k_cvtpk, I_CVTPK)
KERNEL32(k_fmamix, I_FMAMIX)
KERNEL32(k_perm, I_PERM)

// 64-bit register chains: f64 fma, packed f32 fma, 32x32 -> 64 multiply-add
#define CHAIN8D(INS)                                                                                                    \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                  \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")
#define KERNEL64(NAME, INS)                                                                                             \
  __global__ void __launch_bounds__(1024) NAME(float* out, int iters) {                                                 \
    double a0 = threadIdx.x + 1.5, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const double b = 1.0000001, c = 1e-9;                                                                               \
    for (int i = 0; i < iters; ++i) {                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) CHAIN8D(INS);                                                       \
    }                                                                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                        \
  }
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define I_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
KERNEL64(k_fma64, I_FMA64)
KERNEL64(k_add64, I_ADD64)
KERNEL64(k_pkfma, I_PKFMA)
KERNEL64(k_pkmul, I_PKMUL)

// v_mad_u64_u32 vdst[2], sdst[2] (carry), src0, src1, src2[2]: the full 32x32 product in ONE instruction
__global__ void __launch_bounds__(1024) k_mad64(float* out, int iters) {
  unsigned long long a0 = threadIdx.x + 3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const unsigned b = 0xD2511F53u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#define I_MAD(n) "v_mad_u64_u32 %" #n ", vcc, %8, %8, %" #n "\n"
      asm volatile(I_MAD(0) I_MAD(1) I_MAD(2) I_MAD(3) I_MAD(4) I_MAD(5) I_MAD(6) I_MAD(7)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

// f16 MFMA 32x32x16 on 4 independent accumulators (64 accumulator registers)
__global__ void __launch_bounds__(1024) k_mfma(float* out, int iters) {
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// an MFMA followed by independent v_fma_f32: how many vector instructions hide under one matrix instruction
template <int NV>
__global__ void __launch_bounds__(1024) k_mfma_valu(float* out, int iters) {
  f32x16 c0 = {}, c1 = {};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
  float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float bb = 1.0000001f, cc = 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 8; ++v)
        asm volatile(I_FMA(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FMA(5) I_FMA(6) I_FMA(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bb), "v"(cc));
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 8; ++v)
        asm volatile(I_FMA(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FMA(5) I_FMA(6) I_FMA(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bb), "v"(cc));
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// waves with an even index issue only MFMAs, odd ones only v_fma_f32 (NV per MFMA of the partner): do the two pipes of a SIMD overlap
// ACROSS waves? block = 4 SIMDs x 2 waves; wave w sits on SIMD w % 4, so waves w and w + 4 share a SIMD
template <int NV, int ACC_AGPR>
__global__ void __launch_bounds__(512) k_split(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
  float s = 0;
  if (wave < 4) {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (ACC_AGPR) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                       "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                       : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
        } else {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  } else {
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float bb = 1.0000001f, cc = 1e-9f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int v = 0; v < NV; ++v)       // NV groups of 8 = NV vector instructions per partner MFMA (8 MFMAs per iteration)
        asm volatile(I_FMA(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FMA(5) I_FMA(6) I_FMA(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bb), "v"(cc));
    }
    s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same-wave interleave with the accumulators in AGPRs / with 16x16x32 tiles
template <int NV, int SHAPE16>
__global__ void __launch_bounds__(1024) k_mfma_valu_a(float* out, int iters) {
  f32x16 c0 = {}, c1 = {};
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 d0 = {}, d1 = {};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
  float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float bb = 1.0000001f, cc = 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (SHAPE16) { if (u & 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(d1) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(d0) : "v"(a), "v"(b)); }
      else { if (u & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b)); }
#pragma unroll
      for (int v = 0; v < NV / 8; ++v)
        asm volatile(I_FMA(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FMA(5) I_FMA(6) I_FMA(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bb), "v"(cc));
    }
  }
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// NCH dependent chains of v_fma_f32 (1 = every instruction waits for the previous one): the result-to-use latency a wave sees
template <int NCH>
__global__ void __launch_bounds__(1024) k_chain(float* out, int iters) {
  float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  const float b = 1.0000001f, c = 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / NCH; ++u) {
      if (NCH == 1) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));
      if (NCH == 2) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
      if (NCH == 4) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <typename K>
static void run_split(const char* name, K kern, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %7.3f ns per MFMA of the matrix wave (alone: 14.7)\n", name, (double)ms * 1e6 / ((double)iters * 8));
}

template <typename K>
static void run(const char* name, K kern, int per_iter, float* out) {
  printf("%-28s", name);
  for (int wps : {1, 2, 4}) {
    const int threads = 64 * 4 * wps, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %d w/SIMD: %7.3f ns", wps, (double)ms * 1e6 / ((double)iters * per_iter * wps));
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  printf("   (per wave-instruction per SIMD)\n");
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run("v_fma_f32", k_fma, 64, out);
  run("v_fma_f32, 1 dependent chain", k_chain<1>, 64, out);
  run("v_fma_f32, 2 chains", k_chain<2>, 64, out);
  run("v_fma_f32, 4 chains", k_chain<4>, 64, out);
  run("v_xor_b32", k_xor, 64, out);
  run("v_perm_b32", k_perm, 64, out);
  run("v_cvt_pk_f16_f32", k_cvtpk, 64, out);
  run("v_fma_mix_f32", k_fmamix, 64, out);
  run("v_mul_lo_u32", k_mullo, 64, out);
  run("v_mul_hi_u32", k_mulhi, 64, out);
  run("v_mad_u64_u32", k_mad64, 64, out);
  run("v_exp_f32", k_exp, 64, out);
  run("v_rcp_f32", k_rcp, 64, out);
  run("v_log_f32", k_log, 64, out);
  run("v_fma_f64", k_fma64, 64, out);
  run("v_add_f64", k_add64, 64, out);
  run("v_pk_fma_f32", k_pkfma, 64, out);
  run("v_pk_mul_f32", k_pkmul, 64, out);
  run("v_mfma_f32_32x32x16_f16", k_mfma, 16, out);
  run("mfma + 8 v_fma (per group)", k_mfma_valu<8>, 8, out);
  run("mfma + 16 v_fma (per group)", k_mfma_valu<16>, 8, out);
  run("mfma + 32 v_fma (per group)", k_mfma_valu<32>, 8, out);
  run("AGPR acc: mfma + 8 v_fma (per group)", k_mfma_valu_a<8, 0>, 8, out);
  run("AGPR acc: mfma + 16 v_fma (per group)", k_mfma_valu_a<16, 0>, 8, out);
  run("16x16x32 (AGPR) alone (per mfma)", k_mfma_valu_a<0, 1>, 8, out);
  run("16x16x32 (AGPR) + 8 v_fma (per group)", k_mfma_valu_a<8, 1>, 8, out);
  run_split("matrix wave | vector wave, 0 v_fma per mfma", k_split<0, 0>, out);
  run_split("matrix wave | vector wave, 2 v_fma per mfma", k_split<2, 0>, out);
  run_split("matrix wave | vector wave, 4 v_fma per mfma", k_split<4, 0>, out);
  run_split("matrix wave | vector wave, 6 v_fma per mfma", k_split<6, 0>, out);
  run_split("matrix wave | vector wave, 8 v_fma per mfma", k_split<8, 0>, out);
  run_split("same, accumulators in AGPRs, 4 per mfma", k_split<4, 1>, out);
  run_split("same, accumulators in AGPRs, 6 per mfma", k_split<6, 1>, out);
  return 0;
}
