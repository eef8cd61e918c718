// env.hpp — on-device environments: CartPoleEnv{Float32} (RLEnvs 0.6.12 semantics, ppo.jl:82) and the synthetic
// obs-d / reward / done generator used for shapes the reference has no env for (BASELINE config C3).
#pragma once
#include "common.hpp"

namespace crl {

// ------------------------------------------------------------------------------------------------------
// CartPoleEnv{Float32} step (RLEnvs 0.6.12 semantics; oracle/ppo_oracle.c:orc_cartpole_step is the restatement).
// Contraction is off and the promotions to Float64 follow the reference expression (`4 / 3` is a Float64 literal),
// so this is bit-identical to the CPU oracle.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_nt4(f32x4* p, float a, float b, float c, float d) {
  f32x4 v; v[0] = a; v[1] = b; v[2] = c; v[3] = d;
  __builtin_nontemporal_store(v, p);
}

__device__ __forceinline__ float sin_poly(float x) {
  float x2 = x * x;
  float p = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f);
  return __builtin_fmaf(x * x2, p, x);
}
__device__ __forceinline__ float cos_poly(float x) {
  float x2 = x * x;
  float p = __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f);
  return __builtin_fmaf(x2, p, 1.0f);
}

__device__ __forceinline__ bool cartpole_step(float (&s)[4], int& t, int action) {
#pragma clang fp contract(off)
  const float gravity = 9.8f, masspole = 0.1f, totalmass = 1.1f, halflength = 0.5f, pml = 0.05f;
  const float forcemag = 10.0f, dt = 0.02f, ththr = 0.20943951f, xthr = 2.4f;
  t += 1;
  const float force = action == 1 ? forcemag : -forcemag;
  const float xdot = s[1], theta = s[2], thetadot = s[3];
  const float costheta = cos_poly(theta), sintheta = sin_poly(theta);
  const float tmp = (force + (pml * (thetadot * thetadot)) * sintheta) / totalmass;
  const float num = gravity * sintheta - costheta * tmp;
  const double den = (double)halflength * (4.0 / 3.0 - (double)((masspole * (costheta * costheta)) / totalmass));
  const double thetaacc = (double)num / den;
  const double xacc = (double)tmp - (((double)pml * thetaacc) * (double)costheta) / (double)totalmass;
  s[0] = s[0] + dt * xdot;
  s[1] = (float)((double)s[1] + (double)dt * xacc);
  s[2] = s[2] + dt * thetadot;
  s[3] = (float)((double)s[3] + (double)dt * thetaacc);
  return (fabsf(s[0]) > xthr) || (fabsf(s[2]) > ththr) || (t > 500);
}

__device__ __forceinline__ void cartpole_reset(float (&s)[4], uint64_t seed, uint32_t gid, uint64_t gstep, uint32_t stream) {
#pragma clang fp contract(off)
  u32x4 o = philox_env(seed, gid, gstep, stream);
  s[0] = 0.1f * ((float)(o.x >> 8) * 0x1.0p-24f) - 0.05f;
  s[1] = 0.1f * ((float)(o.y >> 8) * 0x1.0p-24f) - 0.05f;
  s[2] = 0.1f * ((float)(o.z >> 8) * 0x1.0p-24f) - 0.05f;
  s[3] = 0.1f * ((float)(o.w >> 8) * 0x1.0p-24f) - 0.05f;
}

// Synthetic env (oracle: synth_step): obs ~ U(-1,1)^d, reward ~ U(-1,1), done ~ Bernoulli(1/200), all from the env's
// own Philox stream (streams 8+q for the observation quads, 3 for reward/done). Stateless: a "reset" is a no-op.
__device__ __forceinline__ float synth_unit(uint32_t o) { return (float)(o >> 8) * 0x1.0p-23f - 1.0f; }
__device__ __forceinline__ void synth_obs4(uint64_t seed, uint32_t gid, uint64_t gstep, int q, float (&o4)[4]) {
  const u32x4 o = philox_env(seed, gid, gstep, 8u + (uint32_t)q);
  o4[0] = synth_unit(o.x); o4[1] = synth_unit(o.y); o4[2] = synth_unit(o.z); o4[3] = synth_unit(o.w);
}
__device__ __forceinline__ void synth_reward_done(uint64_t seed, uint32_t gid, uint64_t gstep, float& reward, bool& done) {
  const u32x4 o = philox_env(seed, gid, gstep, 3u);
  reward = synth_unit(o.x);
  done = (o.y % 200u) == 0u;
}

}  // namespace crl
