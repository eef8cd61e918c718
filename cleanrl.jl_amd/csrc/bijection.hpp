// bijection.hpp — the keyed bijection of [0,n) behind CRL_SHUFFLE_BIJECTION, forward (shuffle.hip) and inverse
// (optim.hip: advantage statistics read the advantages in order and ask which minibatch a sample fell into).
#pragma once
#include "common.hpp"

namespace crl {

struct BijKey { uint32_t k[6]; uint32_t inv[3]; uint32_t mask; int bits; };

// every step is invertible on `bits`-bit words: add key, odd multiply, xorshift, odd multiply, xorshift
__device__ __forceinline__ uint32_t bij_round(uint32_t x, uint32_t mask, int bits, uint32_t k0, uint32_t k1) {
  x = (x + k0) & mask;
  x = (x * (k1 | 1u)) & mask;
  x ^= x >> ((bits + 1) >> 1);
  x = (x * 0x9E3779B1u) & mask;
  x ^= x >> ((bits + 2) / 3);
  return x & mask;
}
__device__ __forceinline__ uint32_t unxorshift(uint32_t x, int s, int bits) {
  for (int sh = s; sh < bits; sh <<= 1) x ^= x >> sh;
  return x;
}
__host__ __device__ __forceinline__ uint32_t inv_odd(uint32_t a) {  // a·inv ≡ 1 (mod 2^32), Newton iteration
  uint32_t x = a;
  for (int i = 0; i < 5; ++i) x *= 2u - a * x;
  return x;
}
__device__ __forceinline__ uint32_t bij_round_inv(uint32_t x, uint32_t mask, int bits, uint32_t k0, uint32_t k1inv) {
  x = unxorshift(x, (bits + 2) / 3, bits);
  x = (x * 0x0E8B2F51u) & mask;                 // 0x9E3779B1^-1 mod 2^32
  x = unxorshift(x, (bits + 1) >> 1, bits);
  x = (x * k1inv) & mask;
  x = (x - k0) & mask;
  return x;
}
__device__ __forceinline__ BijKey bij_key(int n, int bits, uint64_t seed, uint64_t epoch) {
  const u32x4 key = philox(0x51u, (uint32_t)epoch, (uint32_t)(epoch >> 32), 0xB1Du, (uint32_t)seed, (uint32_t)(seed >> 32));
  BijKey b;
  b.k[0] = key.x; b.k[1] = key.y; b.k[2] = key.z; b.k[3] = key.w; b.k[4] = key.y ^ 0xA5A5A5A5u; b.k[5] = key.x ^ 0x3C3C3C3Cu;
  b.inv[0] = inv_odd(b.k[1] | 1u); b.inv[1] = inv_odd(b.k[3] | 1u); b.inv[2] = inv_odd(b.k[5] | 1u);
  b.mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  b.bits = bits;
  (void)n;
  return b;
}
__device__ __forceinline__ uint32_t bij_forward(const BijKey& b, uint32_t p, uint32_t n) {
  uint32_t x = p;
  do {
    x = bij_round(x, b.mask, b.bits, b.k[0], b.k[1]);
    x = bij_round(x, b.mask, b.bits, b.k[2], b.k[3]);
    x = bij_round(x, b.mask, b.bits, b.k[4], b.k[5]);
  } while (x >= n);  // cycle walking keeps it a bijection of [0,n)
  return x;
}
__device__ __forceinline__ uint32_t bij_inverse(const BijKey& b, uint32_t x, uint32_t n) {
  do {
    x = bij_round_inv(x, b.mask, b.bits, b.k[4], b.inv[2]);
    x = bij_round_inv(x, b.mask, b.bits, b.k[2], b.inv[1]);
    x = bij_round_inv(x, b.mask, b.bits, b.k[0], b.inv[0]);
  } while (x >= n);
  return x;
}
__host__ __forceinline__ int bij_bits(long long n) { int bits = 1; while ((1ll << bits) < n) ++bits; return bits; }

}  // namespace crl
