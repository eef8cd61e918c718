// policy.hip — policy/critic forward, categorical sampling, the on-device vectorised CartPole and the persistent
// rollout loop (ppo.jl:21-45,123-166; multi_thread_env.jl:86-133).
//
// One wave owns a tile of 32 envs (the N dimension of v_mfma_f32_32x32x2_f32). Both 64x64 hidden layers run on
// MFMA with the weight A-fragments staged once per launch in LDS and the activations chained register→register
// (the C fragment of one layer is the B operand of the next, no LDS round trip). Envs never interact, so the whole
// `for step in 1:num_steps` loop is ONE launch: 128 dependent steps per wave, no grid synchronisation.
// Stores go to the (·, nt, k) buffer with env fastest: lanes 0-31 write 128/512 contiguous bytes per field per step.
#include <cstdlib>

#include "common.hpp"
#include "env.hpp"
#include "mlp_x3.hpp"
#include "mlp_x2.hpp"
#include "ppo_ctx.hpp"

namespace crl {

// ------------------------------------------------------------------------------------------------------
// get_action / critic on caller-supplied observations (crl_policy_act) — ppo.jl:21-32,128
// ------------------------------------------------------------------------------------------------------
template <int D, int A>
__global__ void __launch_bounds__(256) policy_act_kernel(const float* __restrict__ params, const float* __restrict__ obs,
                                                         const double* __restrict__ u, int n, int32_t* __restrict__ action,
                                                         float* __restrict__ logprob, float* __restrict__ value) {
  using IA = NetImage<D, A, false>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgA = smem;
  float* imgC = smem + IA::SIZE;
  stage_net<D, A, false>(imgA, params, threadIdx.x, blockDim.x);
  stage_net<D, 1, false>(imgC, params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int wpb = blockDim.x >> 6;
  for (int tile = blockIdx.x * wpb + wave; tile * TILE < n; tile += gridDim.x * wpb) {
    const int b = tile * TILE + j;
    const bool ok = b < n;
    float x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = ok ? obs[(size_t)D * b + i] : 0.0f;
    f32x16 h1[2], h2[2];
    float z[A], v[1], p[A], lp[A];
    mlp_forward<D, A, false>(imgA, x, h1, h2, z, lane);
    softmax_logsoftmax<A>(z, p, lp);
    const int a = sample_weights<A>(p, ok ? u[b] : 0.0);
    float lpa = lp[0];
#pragma unroll
    for (int i = 1; i < A; ++i) lpa = (a == i) ? lp[i] : lpa;
    mlp_forward<D, 1, false>(imgC, x, h1, h2, v, lane);
    if (ok && hf == 0) {
      action[b] = a;
      logprob[b] = lpa;
      if (value) value[b] = v[0];
    }
  }
}

// logprob_actions — ppo.jl:34-45 (entropy is the element-wise matrix, Q3)
template <int D, int A>
__global__ void __launch_bounds__(256) logprob_actions_kernel(const float* __restrict__ params, const float* __restrict__ obs,
                                                              const int32_t* __restrict__ actions, int n,
                                                              float* __restrict__ logprob, float* __restrict__ entropy) {
  using IA = NetImage<D, A, false>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  stage_net<D, A, false>(smem, params, threadIdx.x, blockDim.x);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int wpb = blockDim.x >> 6;
  for (int tile = blockIdx.x * wpb + wave; tile * TILE < n; tile += gridDim.x * wpb) {
    const int b = tile * TILE + j;
    const bool ok = b < n;
    float x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = ok ? obs[(size_t)D * b + i] : 0.0f;
    f32x16 h1[2], h2[2];
    float z[A], p[A], lp[A];
    mlp_forward<D, A, false>(smem, x, h1, h2, z, lane);
    softmax_logsoftmax<A>(z, p, lp);
    if (ok && hf == 0) {
      const int a = actions[b];
      float lpa = lp[0];
#pragma unroll
      for (int i = 1; i < A; ++i) lpa = (a == i) ? lp[i] : lpa;
      logprob[b] = lpa;
#pragma unroll
      for (int i = 0; i < A; ++i) entropy[(size_t)A * b + i] = -(p[i] * lp[i]);
    }
  }
  (void)IA::SIZE;
}

// critic(state(env)) for the bootstrap (ppo.jl:169-171; live only in fixed GAE mode, Q10)
template <int D>
__global__ void __launch_bounds__(256) value_kernel(const float* __restrict__ params_critic, const float* __restrict__ obs,
                                                    int n, float* __restrict__ value) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  stage_net<D, 1, false>(smem, params_critic, threadIdx.x, blockDim.x);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int wpb = blockDim.x >> 6;
  for (int tile = blockIdx.x * wpb + wave; tile * TILE < n; tile += gridDim.x * wpb) {
    const int b = tile * TILE + j;
    const bool ok = b < n;
    float x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = ok ? obs[(size_t)D * b + i] : 0.0f;
    f32x16 h1[2], h2[2];
    float v[1];
    mlp_forward<D, 1, false>(smem, x, h1, h2, v, lane);
    if (ok && hf == 0) value[b] = v[0];
  }
}

// reset!(env) at construction + next_obs/next_done initialisation — ppo.jl:80-83,112-115
__global__ void env_reset_kernel(DevCfg c, float* __restrict__ env_state, int32_t* __restrict__ env_t,
                                 float* __restrict__ cur_obs, uint8_t* __restrict__ next_done,
                                 float* __restrict__ ep_return, int32_t* __restrict__ ep_length, double* ep_stats) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) { ep_stats[0] = ep_stats[1] = ep_stats[2] = ep_stats[3] = 0.0; }
  if (e >= c.nt) return;
  float s[4];
  cartpole_reset(s, c.seed, c.env_id_offset + (uint32_t)e, 0, 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) { env_state[4 * e + i] = s[i]; cur_obs[4 * e + i] = s[i]; }
  env_t[e] = 0; next_done[e] = 0; ep_return[e] = 0.0f; ep_length[e] = 0;
}

// ------------------------------------------------------------------------------------------------------
// The rollout loop — ppo.jl:123-166 — one launch for all num_steps.
// ------------------------------------------------------------------------------------------------------
struct RolloutArgs {
  DevCfg c;
  const float* params;
  float* obs; int32_t* action; float* logprob; float* reward; uint8_t* terminal; float* value;
  float* env_state; int32_t* env_t; float* cur_obs; uint8_t* next_done; float* ep_return; int32_t* ep_length;
  double* ep_stats;
  crl_episode_record* ring; uint32_t* ring_count; int ring_cap;   // per-episode records (ring_cap = 0: off)
  uint64_t iteration;
  int stagger;  // s_sleep units (64 clocks) by which waves 4-7 of an 8-wave block start late
  // GAE fused into the tail of the rollout (crl_ppo_iterate, compat mode): the wave that stepped 32 envs for num_steps steps
  // scans their value / reward / terminal columns — which it has just written and which still sit in L2 — backwards and
  // writes advantages and returns (ppo.jl:48-73,173-181): no separate launch, no HBM read of the scan's inputs.
  float* adv; float* ret; int fuse_gae; float gamma, gl;
  double* range_err = nullptr;   // fp16x2 weight-window error word (CX2 kernels)
};

// gae(values, rewards, terminals, γ, λ) for ONE env (this lane), compat mode (ppo.jl:66: the loop starts at k-1, the last slot
// is defined as 0 — Q1): the reference's serial Float64 recurrence, step by step ⇒ bit-identical to orc_gae.
__device__ __forceinline__ void gae_tail_compat(const RolloutArgs& a, int e) {
#pragma clang fp contract(off)
  const int nt = a.c.nt, k = a.c.k;
  size_t idx = (size_t)e + (size_t)nt * (k - 1);
  float vnext = a.value[idx];
  uint32_t tnext = a.terminal[idx];
  a.adv[idx] = 0.0f; a.ret[idx] = 0.0f + vnext;
  double A = 0.0;
  // eight steps' inputs are loaded together (the loads cannot be hoisted above the stores by the compiler: it must assume
  // adv / ret alias the inputs), then the serial recurrence runs on registers: one L2 round trip per eight steps
  constexpr int CH = 8;
  for (int t0 = k - 2; t0 >= 0; t0 -= CH) {
    float v[CH], r[CH]; uint32_t tm[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const bool ok = t0 - i >= 0;
      const size_t ix = ok ? (size_t)e + (size_t)nt * (t0 - i) : (size_t)e;
      v[i] = a.value[ix]; r[i] = a.reward[ix]; tm[i] = a.terminal[ix];
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (t0 - i < 0) break;
      const size_t ix = (size_t)e + (size_t)nt * (t0 - i);
      const double nonterm = 1.0 - (double)(tnext ? 1 : 0);
      const double delta = (double)r[i] + ((double)a.gamma * nonterm) * (double)vnext - (double)v[i];
      const double cc = (double)a.gl * nonterm;
      A = delta + (cc * A);
      const float a32 = (float)A;
      a.adv[ix] = a32; a.ret[ix] = a32 + v[i];
      vnext = v[i]; tnext = tm[i];
    }
  }
}

// CX2: the critic runs as fp16x2 with the exp2-based activation (mlp_x2.hpp) — its output is a value compared at 1e-5, while
// the actor keeps bf16x3 + the reference's rational tanh_fast because its output decides action indices that are bit-compared.
// A critic whose hidden-layer weights leave the fp16 window (|w| >= 255) is restaged and run as bf16x3 by the same launch.
template <int A, bool CX2>
__global__ void __launch_bounds__(512, 2) rollout_cartpole_kernel(RolloutArgs a) {
  constexpr int D = 4;
  constexpr int IASIZE = NetImageX3<D, A, false>::SIZE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgA0 = smem;
  float* imgC0 = smem + IASIZE;
  bool cx2 = CX2;   // block-uniform
  stage_net_x3<D, A, false>(imgA0, a.params, threadIdx.x, blockDim.x);
  if (CX2) {
    static_assert(NetImageX3<D, 1, false>::SIZE >= NetImageX2<D, 1, false>::SIZE, "the bf16x3 critic image (three pieces) is the larger one");
    int* flag = reinterpret_cast<int*>(imgC0 + NetImageX3<D, 1, false>::SIZE);
    if (!stage_net_x2<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x, flag)) {
      if (threadIdx.x == 0 && blockIdx.x == 0) a.range_err[0] = 1.0;   // informational: the fallback ran
      cx2 = false;
    }
  }
  if (!cx2) stage_net_x3<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x);
  __syncthreads();
  const DevCfg& c = a.c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hf = lane >> 5;
  const int wpb = blockDim.x >> 6;
  const int e = (blockIdx.x * wpb + wave) * TILE + j;
  const bool ok = e < c.nt;
  const bool writer = ok && hf == 0;
  const int ee = ok ? e : 0;
  const uint32_t gid = c.env_id_offset + (uint32_t)ee;

  float s[4], co[4];
  {
    const float4 sv = reinterpret_cast<const float4*>(a.env_state)[ee];
    const float4 cv = reinterpret_cast<const float4*>(a.cur_obs)[ee];
    s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
    co[0] = cv.x; co[1] = cv.y; co[2] = cv.z; co[3] = cv.w;
  }
  int t_env = a.env_t[ee];
  uint8_t nd = a.next_done[ee];
  float ep_ret = a.ep_return[ee];
  int ep_len = a.ep_length[ee];
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;

  // Waves w and w+4 of an 8-wave block share a SIMD and run the same program: started together they march through
  // the MFMA-heavy and VALU-only phases of a step in lockstep and leave the matrix pipe idle; a one-time delay of
  // the second half keeps one wave's VALU phase under the other's MFMAs for all 128 steps.
  if (__builtin_amdgcn_readfirstlane(wave) >= 4) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }

  for (int step = 0; step < c.k; ++step) {
    const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
    const size_t b = (size_t)ee + (size_t)c.nt * step;
    ep_len += 1;                                                     // ppo.jl:125
    f32x16 h1[2], h2[2];
    float z[A], v[1], p[A], lp[A];
    // opaque per-step offset: the weight fragments are re-read from LDS every step instead of being hoisted out of
    // the 128-step loop into ~250 registers (which would halve the waves per SIMD)
    int lds_off = 0;
    asm volatile("" : "+v"(lds_off));
    const float* imgA = imgA0 + lds_off;
    const float* imgC = imgC0 + lds_off;
    mlp_forward_x3<D, A, false>(imgA, co, h1, h2, z, lane);          // ppo.jl:127 get_action
    softmax_logsoftmax<A>(z, p, lp);
    const double u = u53(philox_env(c.seed, gid, gstep, 0));
    const int act = sample_weights<A>(p, u);
    float lpa = lp[0];
#pragma unroll
    for (int i = 1; i < A; ++i) lpa = (act == i) ? lp[i] : lpa;
    if (CX2 && cx2) mlp_forward_x2<D, 1, false>(imgC, co, h1, h2, v, lane);  // ppo.jl:128
    else mlp_forward_x3<D, 1, false>(imgC, co, h1, h2, v, lane);
    const bool done = cartpole_step(s, t_env, act);                  // ppo.jl:130
    const float rew = done ? 0.0f : 1.0f;                            // ppo.jl:132 (RLEnvs: reward 0 on the terminal step)
    if (writer) {                                                    // ppo.jl:133-140 Buffer.add!
      // obs/action/logprob are next read by the update pass, a full GAE + shuffle later: stream them past the caches
      // (nontemporal) so the 75 MB the GAE scan needs (value, reward, terminal) stay resident in L2 / Infinity Cache
      store_nt4(reinterpret_cast<f32x4*>(a.obs) + b, co[0], co[1], co[2], co[3]);
      __builtin_nontemporal_store(act, a.action + b); __builtin_nontemporal_store(lpa, a.logprob + b);
      a.reward[b] = rew; a.terminal[b] = nd; a.value[b] = v[0];
    }
    co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3];         // ppo.jl:143 next_obs (before reset!, Q7)
    nd = done ? 1 : 0;                                               // ppo.jl:144
    ep_ret += rew;                                                   // ppo.jl:145
    if (done) {                                                      // ppo.jl:147-165
      if (writer) {
        st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, (double)ep_ret);
        if (a.ring_cap > 0) {
          const uint32_t slot = atomicAdd(a.ring_count, 1u);
          if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
        }
      }
      ep_ret = 0.0f; ep_len = 0;
      cartpole_reset(s, c.seed, gid, gstep, 1);                      // ppo.jl:164 reset!(env)
      t_env = 0;
      if (!c.stale_obs) { co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3]; }
    }
  }
  if (writer) {
    reinterpret_cast<float4*>(a.env_state)[e] = make_float4(s[0], s[1], s[2], s[3]);
    reinterpret_cast<float4*>(a.cur_obs)[e] = make_float4(co[0], co[1], co[2], co[3]);
    a.env_t[e] = t_env; a.next_done[e] = nd; a.ep_return[e] = ep_ret; a.ep_length[e] = ep_len;
  }
  if (a.fuse_gae) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's own stores are what the scan reads back
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (writer) gae_tail_compat(a, e);
  }
  // episode statistics of this rollout ("Episode Statistics" record, aggregated): one atomic set per wave
  st_n = wave_sum(st_n); st_ret = wave_sum(st_ret); st_len = wave_sum(st_len);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) st_max = fmax(st_max, __shfl_xor(st_max, o, 64));
  if (lane == 0 && st_n > 0.0) {
    atomicAdd(&a.ep_stats[0], st_n); atomicAdd(&a.ep_stats[1], st_ret); atomicAdd(&a.ep_stats[2], st_len);
    // return_max: episode returns are non-negative integers ≤ 501 here, so the f64 bit pattern orders like u64
    atomicMax(reinterpret_cast<unsigned long long*>(&a.ep_stats[3]), (unsigned long long)__double_as_longlong(st_max));
  }
}

// Small shards (≤ 512 tiles = 16384 envs, e.g. 8192 envs per GPU under 8-way data parallelism) leave most SIMDs idle and the
// 128 dependent steps set the time. Here each 32-env tile gets TWO waves: wave 0 runs the actor, the sampling and
// the env; wave 1 runs the critic on the same observations, handed over through a double-buffered 512-B LDS slot
// with one s_barrier per step. Step latency drops to the longer of the two halves.
template <int A>
__global__ void __launch_bounds__(128) rollout_split_kernel(RolloutArgs a) {
  constexpr int D = 4;
  constexpr int IASIZE = NetImageX3<D, A, false>::SIZE, ICSIZE = NetImageX3<D, 1, false>::SIZE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgA0 = smem;
  float* imgC0 = smem + IASIZE;
  float4* xch = reinterpret_cast<float4*>(smem + IASIZE + ICSIZE);  // [2][TILE] observations
  stage_net_x3<D, A, false>(imgA0, a.params, threadIdx.x, blockDim.x);
  stage_net_x3<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x);
  const DevCfg& c = a.c;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, hf = lane >> 5;
  const int e = blockIdx.x * TILE + j;
  const bool ok = e < c.nt;
  const bool writer = ok && hf == 0;
  const int ee = ok ? e : 0;
  const uint32_t gid = c.env_id_offset + (uint32_t)ee;

  float s[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0};
  int t_env = 0, ep_len = 0;
  uint8_t nd = 0;
  float ep_ret = 0.0f;
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  if (wave == 0) {
    const float4 sv = reinterpret_cast<const float4*>(a.env_state)[ee];
    const float4 cv = reinterpret_cast<const float4*>(a.cur_obs)[ee];
    s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
    co[0] = cv.x; co[1] = cv.y; co[2] = cv.z; co[3] = cv.w;
    t_env = a.env_t[ee]; nd = a.next_done[ee]; ep_ret = a.ep_return[ee]; ep_len = a.ep_length[ee];
    if (hf == 0) xch[j] = cv;
  }
  __syncthreads();

  for (int step = 0; step < c.k; ++step) {
    const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
    const size_t b = (size_t)ee + (size_t)c.nt * step;
    int lds_off = 0;
    asm volatile("" : "+v"(lds_off));  // keep the weight reads in LDS (see rollout_cartpole_kernel)
    f32x16 h1[2], h2[2];
    if (wave == 0) {
      ep_len += 1;                                                   // ppo.jl:125
      float z[A], p[A], lp[A];
      // the step's uniform does not depend on the network: drawn first, its ≈80 integer instructions can sit in the shadow of
      // the forward pass's MFMA chains instead of behind the softmax on the step's critical path
      const double u = u53(philox_env(c.seed, gid, gstep, 0));
      mlp_forward_x3<D, A, false>(imgA0 + lds_off, co, h1, h2, z, lane);  // ppo.jl:127 get_action
      softmax_logsoftmax<A>(z, p, lp);
      const int act = sample_weights<A>(p, u);
      float lpa = lp[0];
#pragma unroll
      for (int i = 1; i < A; ++i) lpa = (act == i) ? lp[i] : lpa;
      const bool done = cartpole_step(s, t_env, act);                // ppo.jl:130
      const float rew = done ? 0.0f : 1.0f;                          // ppo.jl:132
      if (writer) {                                                  // ppo.jl:133-140 Buffer.add! (value: wave 1)
        store_nt4(reinterpret_cast<f32x4*>(a.obs) + b, co[0], co[1], co[2], co[3]);
        __builtin_nontemporal_store(act, a.action + b); __builtin_nontemporal_store(lpa, a.logprob + b);
        a.reward[b] = rew; a.terminal[b] = nd;
      }
      co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3];       // ppo.jl:143
      nd = done ? 1 : 0;                                             // ppo.jl:144
      ep_ret += rew;                                                 // ppo.jl:145
      if (done) {                                                    // ppo.jl:147-165
        if (writer) {
          st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, (double)ep_ret);
          if (a.ring_cap > 0) {
            const uint32_t slot = atomicAdd(a.ring_count, 1u);
            if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
          }
        }
        ep_ret = 0.0f; ep_len = 0;
        cartpole_reset(s, c.seed, gid, gstep, 1);                    // ppo.jl:164
        t_env = 0;
        if (!c.stale_obs) { co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3]; }
      }
      if (hf == 0) xch[((step + 1) & 1) * TILE + j] = make_float4(co[0], co[1], co[2], co[3]);
    } else {
      const float4 cv = xch[(step & 1) * TILE + j];
      const float cx[4] = {cv.x, cv.y, cv.z, cv.w};
      float v[1];
      mlp_forward_x3<D, 1, false>(imgC0 + lds_off, cx, h1, h2, v, lane);  // ppo.jl:128
      if (writer) a.value[b] = v[0];
    }
    __syncthreads();
  }
  if (wave == 0) {
    if (writer) {
      reinterpret_cast<float4*>(a.env_state)[e] = make_float4(s[0], s[1], s[2], s[3]);
      reinterpret_cast<float4*>(a.cur_obs)[e] = make_float4(co[0], co[1], co[2], co[3]);
      a.env_t[e] = t_env; a.next_done[e] = nd; a.ep_return[e] = ep_ret; a.ep_length[e] = ep_len;
      // value[] was written by wave 1 of this block; the step loop's closing __syncthreads() made it visible here
      if (a.fuse_gae) gae_tail_compat(a, e);
    }
    st_n = wave_sum(st_n); st_ret = wave_sum(st_ret); st_len = wave_sum(st_len);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) st_max = fmax(st_max, __shfl_xor(st_max, o, 64));
    if (lane == 0 && st_n > 0.0) {
      atomicAdd(&a.ep_stats[0], st_n); atomicAdd(&a.ep_stats[1], st_ret); atomicAdd(&a.ep_stats[2], st_len);
      atomicMax(reinterpret_cast<unsigned long long*>(&a.ep_stats[3]), (unsigned long long)__double_as_longlong(st_max));
    }
  }
}

// The same shards on THREE waves per tile (fp16x2 builds): the 128 dependent steps are what a small shard's rollout costs (0.51 ms
// at any size up to 16384 envs), and a lone wave issues one instruction every ≈5 cycles, so the step time is the actor wave's
// instruction count. Waves 0 and 1 each compute HALF the actor's hidden rows (rows 32·w … 32·w+31 of both layers): half the
// tanh_fast evaluations, half the bf16 splits, one 24-MFMA chain instead of two. What the halves owe each other goes through LDS:
// the split pieces of h1 (each wave's rows are two of the four k-steps of the other's layer-2 product) and the head's dot
// product, which stays ONE chain over the 64 rows in mlp_forward_x3's order — wave 0 runs its 16 terms per lane half, wave 1
// continues from that partial, adds the halves and the bias — so every logit is bit-identical to the one-wave forward and the
// sampled actions do not change. Wave 2 runs the critic as fp16x2 with the exp2 activation (as rollout_cartpole_kernel<…, CX2>),
// cut in two so that it is never the last wave at a barrier. Wave 1, which ends the head's chain and so has the logits first, also samples and owns the
// env (the logits never travel). Three block barriers per step.
template <int A>
__global__ void __launch_bounds__(192) rollout_split3_kernel(RolloutArgs a) {
  constexpr int D = 4;
  using IA = NetImageX3<D, A, false>;
  using IC = NetImageX2<D, 1, false>;
  constexpr int ICMAX = NetImageX3<D, 1, false>::SIZE;   // room for the bf16x3 fallback image of the critic
  static_assert(ICMAX >= IC::SIZE, "the bf16x3 critic image (three pieces) is the larger one");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgA0 = smem;
  float* imgC0 = imgA0 + IA::SIZE;
  float4* xch = reinterpret_cast<float4*>(imgC0 + ICMAX);                  // [2][TILE] observations
  bf16x8* pcs = reinterpret_cast<bf16x8*>(reinterpret_cast<float*>(xch) + 2 * TILE * 4);   // [2 waves][2 k-steps][3 pieces][64 lanes]
  float* hd = reinterpret_cast<float*>(pcs + 2 * 2 * 3 * 64);              // [A][64] wave 0's partial head sums
  int* flag = reinterpret_cast<int*>(hd + A * 64);
  stage_net_x3<D, A, false>(imgA0, a.params, threadIdx.x, blockDim.x);
  bool cx2 = true;   // block-uniform: false = the critic's weights left the fp16 window (|w| >= 255) and it runs as bf16x3
  if (!stage_net_x2<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x, flag)) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.range_err[0] = 1.0;   // informational: the fallback ran
    cx2 = false;
    stage_net_x3<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x);
  }
  const DevCfg& c = a.c;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, hf = lane >> 5;
  const int e = blockIdx.x * TILE + j;
  const bool ok = e < c.nt;
  const bool writer = ok && hf == 0;
  const int ee = ok ? e : 0;
  const uint32_t gid = c.env_id_offset + (uint32_t)ee;

  float s[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0};
  int t_env = 0, ep_len = 0;
  uint8_t nd = 0;
  float ep_ret = 0.0f;
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  if (wave == 1) {   // the wave that ends the head's chain has the logits first: it samples and owns the env
    const float4 sv = reinterpret_cast<const float4*>(a.env_state)[ee];
    const float4 cv = reinterpret_cast<const float4*>(a.cur_obs)[ee];
    s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
    co[0] = cv.x; co[1] = cv.y; co[2] = cv.z; co[3] = cv.w;
    t_env = a.env_t[ee]; nd = a.next_done[ee]; ep_ret = a.ep_return[ee]; ep_len = a.ep_length[ee];
    if (hf == 0) xch[j] = cv;
  }
  __syncthreads();

  for (int step = 0; step < c.k; ++step) {
    const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
    const size_t b = (size_t)ee + (size_t)c.nt * step;
    int lds_off = 0;
    asm volatile("" : "+v"(lds_off));  // keep the weight reads in LDS (see rollout_cartpole_kernel)
    if (wave < 2) {
      const int mo = wave;
      const float* img = imgA0 + lds_off;
      float cx[4];
      if (wave == 1) { cx[0] = co[0]; cx[1] = co[1]; cx[2] = co[2]; cx[3] = co[3]; }
      else { const float4 cv = xch[(step & 1) * TILE + j]; cx[0] = cv.x; cx[1] = cv.y; cx[2] = cv.z; cx[3] = cv.w; }
      double u = 0.0;
      if (wave == 1) { ep_len += 1; u = u53(philox_env(c.seed, gid, gstep, 0)); }           // ppo.jl:125; drawn early (see split kernel)
      // layer 1, this wave's 32 rows
      f32x16 acc = load16(img + IA::B1C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < D / 2; ++ks) {
        const float bv = hf ? cx[2 * ks + 1] : cx[2 * ks];
        acc = mfma32(img[IA::WF1 + (mo * (D / 2) + ks) * 64 + lane], bv, acc);
      }
      f32x16 h;
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = tanh_fast(acc[r]);
      // my two k-steps of the layer-2 product (k-step 2·mo + q = registers 8q … 8q+7), split once, shared through LDS
      P3 mine[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float xb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xb[i] = h[8 * q + i];
        mine[q] = split3(xb);
        bf16x8* dst = pcs + ((mo * 2 + q) * 3) * 64 + lane;
        dst[0] = mine[q].hi; dst[64] = mine[q].mid; dst[128] = mine[q].lo;
      }
      __syncthreads();                                                                       // (1) both halves of h1 are published
      P3 other[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const bf16x8* src = pcs + (((1 - mo) * 2 + q) * 3) * 64 + lane;
        other[q].hi = src[0]; other[q].mid = src[64]; other[q].lo = src[128];
      }
      acc = load16(img + IA::B2C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const P3& bq = ((ks >> 1) == mo) ? mine[ks & 1] : other[ks & 1];
        acc = mfma_x3(load_wfrag(img + IA::WF2P, mo, ks, lane), bq, acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) h[r] = tanh_fast(acc[r]);
      // head: one chain per lane half over idx = 0 … 31 (mlp_forward_x3): wave 0 owns idx 0-15, wave 1 continues with 16-31
      if (mo == 0) {
#pragma unroll
        for (int o = 0; o < A; ++o) {
          const f32x4* w = reinterpret_cast<const f32x4*>(img + IA::W3 + o * 64 + hf * 32);
          float accv = 0.0f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 wv = w[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) accv = __builtin_fmaf(wv[i], h[q * 4 + i], accv);
          }
          hd[o * 64 + lane] = accv;
        }
      }
      __syncthreads();                                                                       // (2) wave 0's partial sums are there
      if (mo == 1) {
        float z[A], p[A], lp[A];
#pragma unroll
        for (int o = 0; o < A; ++o) {
          const f32x4* w = reinterpret_cast<const f32x4*>(img + IA::W3 + o * 64 + hf * 32);
          float accv = hd[o * 64 + lane];
#pragma unroll
          for (int q = 4; q < 8; ++q) {
            const f32x4 wv = w[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) accv = __builtin_fmaf(wv[i], h[(q - 4) * 4 + i], accv);
          }
          z[o] = add32(accv) + img[IA::B3 + o];
        }
        softmax_logsoftmax<A>(z, p, lp);                                                     // ppo.jl:127 get_action
        const int act = sample_weights<A>(p, u);
        float lpa = lp[0];
#pragma unroll
        for (int i = 1; i < A; ++i) lpa = (act == i) ? lp[i] : lpa;
        const bool done = cartpole_step(s, t_env, act);                                      // ppo.jl:130
        const float rew = done ? 0.0f : 1.0f;                                                // ppo.jl:132
        if (writer) {                                                                        // ppo.jl:133-140 Buffer.add! (value: wave 2)
          store_nt4(reinterpret_cast<f32x4*>(a.obs) + b, co[0], co[1], co[2], co[3]);
          __builtin_nontemporal_store(act, a.action + b); __builtin_nontemporal_store(lpa, a.logprob + b);
          a.reward[b] = rew; a.terminal[b] = nd;
        }
        co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3];                             // ppo.jl:143
        nd = done ? 1 : 0;                                                                   // ppo.jl:144
        ep_ret += rew;                                                                       // ppo.jl:145
        if (done) {                                                                          // ppo.jl:147-165
          if (writer) {
            st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, (double)ep_ret);
            if (a.ring_cap > 0) {
              const uint32_t slot = atomicAdd(a.ring_count, 1u);
              if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
            }
          }
          ep_ret = 0.0f; ep_len = 0;
          cartpole_reset(s, c.seed, gid, gstep, 1);                                          // ppo.jl:164
          t_env = 0;
          if (!c.stale_obs) { co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3]; }
        }
        if (hf == 0) xch[((step + 1) & 1) * TILE + j] = make_float4(co[0], co[1], co[2], co[3]);
      }
    } else if (!cx2) {
      // critic as bf16x3 (the fallback flavour): same three barriers per step
      const float4 cv = xch[(step & 1) * TILE + j];
      const float cx[4] = {cv.x, cv.y, cv.z, cv.w};
      f32x16 h1[2], h2[2];
      float v[1];
      mlp_forward_x3<D, 1, false>(imgC0 + lds_off, cx, h1, h2, v, lane);                     // ppo.jl:128
      __syncthreads();                                                                       // (1)
      __syncthreads();                                                                       // (2)
      if (writer) a.value[b] = v[0];
    } else {
      // critic (ppo.jl:128): mlp_forward_x2 cut at the layer boundary so that each part is shorter than the actor's segment it faces
      const float* img = imgC0 + lds_off;
      const float4 cv = xch[(step & 1) * TILE + j];
      const float cx[4] = {cv.x, cv.y, cv.z, cv.w};
      f32x16 a0 = load16(img + IC::B1C + hf * 32), a1 = load16(img + IC::B1C + hf * 32 + 16);
#pragma unroll
      for (int ks = 0; ks < D / 2; ++ks) {
        const float bv = hf ? cx[2 * ks + 1] : cx[2 * ks];
        a0 = mfma32(img[IC::WF1 + (0 * (D / 2) + ks) * 64 + lane], bv, a0);
        a1 = mfma32(img[IC::WF1 + (1 * (D / 2) + ks) * 64 + lane], bv, a1);
      }
      f32x16 h1s[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { h1s[0][r] = tanh_exp2_arg(a0[r], X2_ACT_SCALE); h1s[1][r] = tanh_exp2_arg(a1[r], X2_ACT_SCALE); }
      __syncthreads();                                                                       // (1)
      a0 = load16(img + IC::B2C + hf * 32); a1 = load16(img + IC::B2C + hf * 32 + 16);
      dense64_x2(img + IC::WF2H, h1s, a0, a1, lane);
      const f32x4* w = reinterpret_cast<const f32x4*>(img + IC::W3 + hf * 32);
      float accv = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 wv = w[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = q * 4 + i;
          const float hv = tanh_exp2((idx >> 4) ? a1[idx & 15] : a0[idx & 15], TWO_LOG2E * X2_FWD_UNSCALE, 1.0f);
          accv = __builtin_fmaf(wv[i], hv, accv);
        }
      }
      const float v = add32(accv) + img[IC::B3];
      __syncthreads();                                                                       // (2)
      if (writer) a.value[b] = v;
    }
    __syncthreads();                                                                         // (3) next observations are published
  }
  if (wave == 1) {
    if (writer) {
      reinterpret_cast<float4*>(a.env_state)[e] = make_float4(s[0], s[1], s[2], s[3]);
      reinterpret_cast<float4*>(a.cur_obs)[e] = make_float4(co[0], co[1], co[2], co[3]);
      a.env_t[e] = t_env; a.next_done[e] = nd; a.ep_return[e] = ep_ret; a.ep_length[e] = ep_len;
      // value[] was written by wave 2 of this block; the step loop's closing __syncthreads() made it visible here
      if (a.fuse_gae) gae_tail_compat(a, e);
    }
    st_n = wave_sum(st_n); st_ret = wave_sum(st_ret); st_len = wave_sum(st_len);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) st_max = fmax(st_max, __shfl_xor(st_max, o, 64));
    if (lane == 0 && st_n > 0.0) {
      atomicAdd(&a.ep_stats[0], st_n); atomicAdd(&a.ep_stats[1], st_ret); atomicAdd(&a.ep_stats[2], st_len);
      atomicMax(reinterpret_cast<unsigned long long*>(&a.ep_stats[3]), (unsigned long long)__double_as_longlong(st_max));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same shards on SIX waves per tile (option rollout_split = 3; verdict r5 item 4 ii). What a small shard's rollout costs is the latency of one step
// (128 dependent steps; 3.5 µs each in rollout_split3_kernel), and a lone wave issues an instruction every ≈5 cycles — so a step is as long as the
// longest wave's instruction stream plus its matrix chain. Here the actor's 64 hidden rows go over FOUR waves (16 rows each: 8 tanh_fast and 8 split
// elements per lane and layer instead of 16) and its 64x64 product runs on v_mfma_f32_16x16x32_bf16 — two independent 12-deep chains of short
// instructions per wave instead of one 24-deep chain of v_mfma_f32_32x32x16_bf16 — and the critic's rows over TWO waves (fp16x2, exp2 activation; the
// split3 kernel's actor structure). Operands cross waves through LDS in the consumer's fragment order:
//   * C layout of the 16x16 tiles: lane (n = l & 15, g = l >> 4), register i = row 4g + i of the wave's 16 rows, one tile per 16 samples (ct = 0, 1);
//   * B fragment of k-step s (32 of the 64 h1 rows), lane (n, g), slot j = row 16·(2s + (j >> 2)) + 4g + (j & 3): four rows from wave 2s, four from
//     wave 2s + 1, both in the SAME lane's registers — a producer stores 8 bytes per piece, a consumer loads 16 (update16.hpp's kmap16 order);
//   * the weight A-fragments are staged in that k order.
// The head is a plain sum here (each wave its 16 rows, folded over row groups and waves): the 16x16x32 product already sums a logit's terms in another order
// than the one-wave forward, so logits agree with it — and with the oracle — to float32 rounding, not bit for bit; actions then differ from the
// oracle's only where the draw sits within rounding of a CDF knot, which is the rule every rollout kernel is tested under (tests/test_gpu_parity.py).
// Wave 0, lanes 0-31, samples and owns the envs. Three block barriers per step, as in the three-wave kernel.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float rows4_sum_f(float v) {     // v + the same lane's value in the three other 16-lane rows of the wave
  const unsigned x = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned y = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(y, y, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int A>
struct NetImageA16 {   // actor image of the six-wave kernel, offsets in floats
  static constexpr int WF2 = 0;                       // [piece 3][t 4][s 2][lane 64][8 bf16]: W2[16t + (lane & 15)][16(2s + (j >> 2)) + 4(lane >> 4) + (j & 3)]
  static constexpr int WF1 = WF2 + 3 * 4 * 2 * 64 * 4;   // [t 4][lane 64]: W1[16t + (lane & 15)][lane >> 4]
  static constexpr int B1 = WF1 + 256;                // plain [64]
  static constexpr int B2 = B1 + 64;
  static constexpr int W3 = B2 + 64;                  // [a][64]
  static constexpr int B3 = W3 + A * 64;
  static constexpr int SIZE = ((B3 + A + 3) / 4) * 4;
};
template <int A>
__device__ __forceinline__ void stage_net_a16(float* img, const float* __restrict__ p, int tid, int nthreads) {
  using I = NetImageA16<A>;
  using P = NetParams<4, A>;
  __bf16* wf = reinterpret_cast<__bf16*>(img + I::WF2);
  for (int idx = tid; idx < 4096; idx += nthreads) {
    const int j = idx & 7, lane = (idx >> 3) & 63, s2 = (idx >> 9) & 1, t = idx >> 10;
    const int row = 16 * t + (lane & 15), k = 16 * (2 * s2 + (j >> 2)) + 4 * (lane >> 4) + (j & 3);
    const float w = p[P::W2 + row + H * k];
    const __bf16 h = (__bf16)w; const float r1 = w - (float)h;
    const __bf16 m = (__bf16)r1; const float r2 = r1 - (float)m;
    wf[idx] = h; wf[4096 + idx] = m; wf[8192 + idx] = (__bf16)r2;
  }
  for (int idx = tid; idx < 256; idx += nthreads) img[I::WF1 + idx] = p[P::W1 + (16 * (idx >> 6) + (idx & 15)) + H * ((idx & 63) >> 4)];
  for (int idx = tid; idx < 64; idx += nthreads) { img[I::B1 + idx] = p[P::B1 + idx]; img[I::B2 + idx] = p[P::B2 + idx]; }
  for (int idx = tid; idx < A * 64; idx += nthreads) img[I::W3 + idx] = p[P::W3 + (idx >> 6) + A * (idx & 63)];
  for (int idx = tid; idx < A; idx += nthreads) img[I::B3 + idx] = p[P::B3 + idx];
}

// 128 VGPRs at most: six waves on four SIMDs put two of them on two SIMDs, and 2 x 128 leaves the other half of those register files to the shuffle's leaf blocks
// (1024 threads = four 56-register waves per SIMD), which otherwise wait for the whole rollout to end and the iteration's first leg gains nothing
template <int A>
__global__ void __attribute__((amdgpu_flat_work_group_size(384, 384), amdgpu_waves_per_eu(4, 4))) rollout_split6_kernel(RolloutArgs a) {
  constexpr int D = 4;
  using IA = NetImageA16<A>;
  using IC = NetImageX2<D, 1, false>;
  constexpr int ICMAX = NetImageX3<D, 1, false>::SIZE;   // room for the bf16x3 fallback image of the critic
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgA0 = smem;
  float* imgC0 = imgA0 + IA::SIZE;
  float* xch = imgC0 + ICMAX;                                              // [2][TILE][4] observations
  uint2* pcsA = reinterpret_cast<uint2*>(xch + 2 * TILE * 4);              // [s 2][ct 2][piece 3][lane 64][half 2] x 8 B: the actor's h1 pieces in B-fragment order
  f16x8* pcsC = reinterpret_cast<f16x8*>(pcsA + 2 * 2 * 3 * 64 * 2);       // [mo 2][q 2][piece 2][lane 64]: the critic's h1 pieces (three bf16 pieces in the fallback flavour: sized for those)
  float* hdA = reinterpret_cast<float*>(pcsC + 2 * 2 * 3 * 64);            // [wave 4][A][TILE] partial logits
  float* hdC = hdA + 4 * A * TILE;                                         // [mo 2][TILE] partial values
  // The two Philox calls of a step — the action draw and the reset state of an env that ends — depend on (seed, env, step) only: waves 1 and 2, idle
  // between barriers (2) and (3) while wave 0 samples and steps the envs, compute them for the NEXT step and leave them here (100 instructions each
  // off the critical wave's path: 2.65 -> 2.3 µs per step)
  double* ubuf = reinterpret_cast<double*>(hdC + 2 * TILE);                // [2][TILE]
  float4* rbuf = reinterpret_cast<float4*>(ubuf + 2 * TILE);               // [2][TILE]
  int* flag = reinterpret_cast<int*>(rbuf + 2 * TILE);
  stage_net_a16<A>(imgA0, a.params, threadIdx.x, blockDim.x);
  bool cx2 = true;   // block-uniform: false = the critic's weights left the fp16 window and wave 4 runs it as bf16x3
  if (!stage_net_x2<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x, flag)) {
    if (threadIdx.x == 0 && blockIdx.x == 0) a.range_err[0] = 1.0;
    cx2 = false;
    stage_net_x3<D, 1, false>(imgC0, a.params + NetParams<D, A>::SIZE, threadIdx.x, blockDim.x);
  }
  const DevCfg& c = a.c;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // env role: wave 0, lanes 0-31, one env each
  const int e = blockIdx.x * TILE + (lane & 31);
  const bool envlane = wave == 0 && lane < 32;
  const bool ok = e < c.nt;
  const bool writer = ok && envlane;
  const int ee = ok ? e : 0;
  const uint32_t gid = c.env_id_offset + (uint32_t)ee;
  float s[4] = {0, 0, 0, 0}, co[4] = {0, 0, 0, 0};
  int t_env = 0, ep_len = 0;
  uint8_t nd = 0;
  float ep_ret = 0.0f;
  double st_n = 0.0, st_ret = 0.0, st_len = 0.0, st_max = 0.0;
  if (envlane) {
    const float4 sv = reinterpret_cast<const float4*>(a.env_state)[ee];
    const float4 cv = reinterpret_cast<const float4*>(a.cur_obs)[ee];
    s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
    co[0] = cv.x; co[1] = cv.y; co[2] = cv.z; co[3] = cv.w;
    t_env = a.env_t[ee]; nd = a.next_done[ee]; ep_ret = a.ep_return[ee]; ep_len = a.ep_length[ee];
    reinterpret_cast<float4*>(xch)[lane] = cv;
  }
  {
    const uint64_t g0 = a.iteration * (uint64_t)c.k;
    if (wave == 1 && lane < 32) ubuf[lane] = u53(philox_env(c.seed, gid, g0, 0));
    if (wave == 2 && lane < 32) { float r[4]; cartpole_reset(r, c.seed, gid, g0, 1); rbuf[lane] = make_float4(r[0], r[1], r[2], r[3]); }
  }
  // actor role (waves 0-3): lane (n, g), rows 16·wave + 4g + i
  const int n = lane & 15, g = lane >> 4;
  float w1a = 0.0f, b1a[4] = {0, 0, 0, 0}, b2a[4] = {0, 0, 0, 0}, w3a[A][4];
#pragma unroll
  for (int o = 0; o < A; ++o)
#pragma unroll
    for (int i = 0; i < 4; ++i) w3a[o][i] = 0.0f;
  __syncthreads();
  if (wave < 4) {
    w1a = imgA0[IA::WF1 + wave * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * wave + 4 * g + i;
      b1a[i] = imgA0[IA::B1 + row]; b2a[i] = imgA0[IA::B2 + row];
#pragma unroll
      for (int o = 0; o < A; ++o) w3a[o][i] = imgA0[IA::W3 + o * 64 + row];
    }
  }

  for (int step = 0; step < c.k; ++step) {
    const uint64_t gstep = a.iteration * (uint64_t)c.k + (uint64_t)step;
    const size_t b = (size_t)ee + (size_t)c.nt * step;
    int lds_off = 0;
    asm volatile("" : "+v"(lds_off));  // keep the weight reads in LDS (see rollout_cartpole_kernel)
    const float* xcur = xch + (step & 1) * (TILE * 4) + lds_off;
    if (wave < 4) {
      if (envlane) ep_len += 1;                                                               // ppo.jl:125
      // layer 1: this wave's 16 rows, two 16-sample tiles
      f32x4 h1[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 acc; acc[0] = b1a[0]; acc[1] = b1a[1]; acc[2] = b1a[2]; acc[3] = b1a[3];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a, xcur[(16 * ct + n) * 4 + g], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) h1[ct][i] = tanh_fast(acc[i]);
      }
      {
        const float xb[8] = {h1[0][0], h1[0][1], h1[0][2], h1[0][3], h1[1][0], h1[1][1], h1[1][2], h1[1][3]};
        const P3 p = split3(xb);
        const u32x4v ph = __builtin_bit_cast(u32x4v, p.hi), pm = __builtin_bit_cast(u32x4v, p.mid), pl = __builtin_bit_cast(u32x4v, p.lo);
        const int s2 = wave >> 1, half = wave & 1;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          uint2* dst = pcsA + ((((s2 * 2 + ct) * 3) * 64 + lane) * 2 + half);
          dst[0] = make_uint2(ph[2 * ct], ph[2 * ct + 1]);
          dst[128] = make_uint2(pm[2 * ct], pm[2 * ct + 1]);
          dst[256] = make_uint2(pl[2 * ct], pl[2 * ct + 1]);
        }
      }
      __syncthreads();                                                                        // (1) all of h1 is published
      f32x4 acc2[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) { acc2[ct][0] = b2a[0]; acc2[ct][1] = b2a[1]; acc2[ct][2] = b2a[2]; acc2[ct][3] = b2a[3]; }
      const bf16x8* wf = reinterpret_cast<const bf16x8*>(imgA0 + IA::WF2 + lds_off);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        P3 wa;
        wa.hi = wf[((0 * 4 + wave) * 2 + s2) * 64 + lane]; wa.mid = wf[((1 * 4 + wave) * 2 + s2) * 64 + lane]; wa.lo = wf[((2 * 4 + wave) * 2 + s2) * 64 + lane];
        P3 bq[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const bf16x8* src = reinterpret_cast<const bf16x8*>(pcsA) + ((s2 * 2 + ct) * 3) * 64 + lane;
          bq[ct].hi = src[0]; bq[ct].mid = src[64]; bq[ct].lo = src[128];
        }
        // the two tiles' chains interleaved, smallest partial products first (mfma_x3's order)
        acc2[0] = mfma16_bf16(wa.lo, bq[0].hi, acc2[0]);   acc2[1] = mfma16_bf16(wa.lo, bq[1].hi, acc2[1]);
        acc2[0] = mfma16_bf16(wa.hi, bq[0].lo, acc2[0]);   acc2[1] = mfma16_bf16(wa.hi, bq[1].lo, acc2[1]);
        acc2[0] = mfma16_bf16(wa.mid, bq[0].mid, acc2[0]); acc2[1] = mfma16_bf16(wa.mid, bq[1].mid, acc2[1]);
        acc2[0] = mfma16_bf16(wa.mid, bq[0].hi, acc2[0]);  acc2[1] = mfma16_bf16(wa.mid, bq[1].hi, acc2[1]);
        acc2[0] = mfma16_bf16(wa.hi, bq[0].mid, acc2[0]);  acc2[1] = mfma16_bf16(wa.hi, bq[1].mid, acc2[1]);
        acc2[0] = mfma16_bf16(wa.hi, bq[0].hi, acc2[0]);   acc2[1] = mfma16_bf16(wa.hi, bq[1].hi, acc2[1]);
      }
      // head partials of this wave's 16 rows: per sample, folded over the four row groups
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float h2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h2[i] = tanh_fast(acc2[ct][i]);
#pragma unroll
        for (int o = 0; o < A; ++o) {
          float pz = w3a[o][0] * h2[0];
#pragma unroll
          for (int i = 1; i < 4; ++i) pz = __builtin_fmaf(w3a[o][i], h2[i], pz);
          pz = rows4_sum_f(pz);
          if (g == 0) hdA[(wave * A + o) * TILE + 16 * ct + n] = pz;
        }
      }
      __syncthreads();                                                                        // (2) every wave's partial logits (and the critic's partial values) are there
      if (envlane) {
        float z[A], p[A], lp[A];
#pragma unroll
        for (int o = 0; o < A; ++o)
          z[o] = ((hdA[(0 * A + o) * TILE + lane] + hdA[(1 * A + o) * TILE + lane]) + (hdA[(2 * A + o) * TILE + lane] + hdA[(3 * A + o) * TILE + lane])) + imgA0[IA::B3 + o];
        softmax_logsoftmax<A>(z, p, lp);                                                      // ppo.jl:127 get_action
        const int act = sample_weights<A>(p, ubuf[(step & 1) * TILE + lane]);                 // the draw of (env, step): Philox stream 0, computed a step ahead by wave 1
        float lpa = lp[0];
#pragma unroll
        for (int i = 1; i < A; ++i) lpa = (act == i) ? lp[i] : lpa;
        const bool done = cartpole_step(s, t_env, act);                                       // ppo.jl:130
        const float rew = done ? 0.0f : 1.0f;                                                 // ppo.jl:132
        if (writer) {                                                                         // ppo.jl:133-140 Buffer.add! (value: wave 4)
          store_nt4(reinterpret_cast<f32x4*>(a.obs) + b, co[0], co[1], co[2], co[3]);
          __builtin_nontemporal_store(act, a.action + b); __builtin_nontemporal_store(lpa, a.logprob + b);
          a.reward[b] = rew; a.terminal[b] = nd;
        }
        co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3];                              // ppo.jl:143
        nd = done ? 1 : 0;                                                                    // ppo.jl:144
        ep_ret += rew;                                                                        // ppo.jl:145
        if (done) {                                                                           // ppo.jl:147-165
          if (writer) {
            st_n += 1.0; st_ret += (double)ep_ret; st_len += (double)ep_len; st_max = fmax(st_max, (double)ep_ret);
            if (a.ring_cap > 0) {
              const uint32_t slot = atomicAdd(a.ring_count, 1u);
              if (slot < (uint32_t)a.ring_cap) a.ring[slot] = crl_episode_record{ep_ret, ep_len, (int32_t)gid, step};
            }
          }
          ep_ret = 0.0f; ep_len = 0;
          { const float4 rv = rbuf[(step & 1) * TILE + lane]; s[0] = rv.x; s[1] = rv.y; s[2] = rv.z; s[3] = rv.w; }   // ppo.jl:164 reset!: cartpole_reset(seed, env, step), computed a step ahead by wave 2
          t_env = 0;
          if (!c.stale_obs) { co[0] = s[0]; co[1] = s[1]; co[2] = s[2]; co[3] = s[3]; }
        }
        reinterpret_cast<float4*>(xch + ((step + 1) & 1) * (TILE * 4))[lane] = make_float4(co[0], co[1], co[2], co[3]);
      } else if (wave == 1 && lane < 32) {
        ubuf[((step + 1) & 1) * TILE + lane] = u53(philox_env(c.seed, gid, gstep + 1, 0));
      } else if (wave == 2 && lane < 32) {
        float r[4];
        cartpole_reset(r, c.seed, gid, gstep + 1, 1);
        rbuf[((step + 1) & 1) * TILE + lane] = make_float4(r[0], r[1], r[2], r[3]);
      }
    } else if (!cx2) {
      // critic as bf16x3 (the fallback flavour: a hidden-layer weight left the fp16 window), in the same two-wave form — wave 4 + mo owns hidden rows 32·mo …
      // (mlp_forward_x3 whole on one wave needs 221 VGPRs, which alone made this kernel too fat for the shuffle's blocks to share its CUs: the iteration's
      // first leg then ended with the shuffle, not the rollout)
      const int mo = wave - 4, j = lane & 31, hf = lane >> 5;
      using IC3 = NetImageX3<D, 1, false>;
      const float* img = imgC0 + lds_off;
      const float4 cv = reinterpret_cast<const float4*>(xcur)[j];
      const float cx[4] = {cv.x, cv.y, cv.z, cv.w};
      f32x16 acc = load16(img + IC3::B1C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < D / 2; ++ks) {
        const float bv = hf ? cx[2 * ks + 1] : cx[2 * ks];
        acc = mfma32(img[IC3::WF1 + (mo * (D / 2) + ks) * 64 + lane], bv, acc);
      }
      bf16x8* pcs3 = reinterpret_cast<bf16x8*>(pcsC);                  // [mo 2][q 2][piece 3][lane 64]
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float xb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xb[i] = tanh_fast(acc[8 * q + i]);
        const P3 mine = split3(xb);
        bf16x8* dst = pcs3 + ((mo * 2 + q) * 3) * 64 + lane;
        dst[0] = mine.hi; dst[64] = mine.mid; dst[128] = mine.lo;
      }
      __syncthreads();                                                                        // (1)
      acc = load16(img + IC3::B2C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {     // every k-step's B pieces come back from LDS (this wave's own too): 12 registers live instead of 48
        const bf16x8* src = pcs3 + (((ks >> 1) * 2 + (ks & 1)) * 3) * 64 + lane;
        P3 bq; bq.hi = src[0]; bq.mid = src[64]; bq.lo = src[128];
        acc = mfma_x3(load_wfrag(img + IC3::WF2P, mo, ks, lane), bq, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      const f32x4* w = reinterpret_cast<const f32x4*>(img + IC3::W3 + hf * 32 + 16 * mo);
      float accv = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = w[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) accv = __builtin_fmaf(wv[i], tanh_fast(acc[q * 4 + i]), accv);
      }
      accv = add32(accv);
      if (hf == 0) hdC[mo * TILE + j] = accv;
      __syncthreads();                                                                        // (2)
      if (mo == 0 && hf == 0 && ok) a.value[(size_t)(blockIdx.x * TILE + j) + (size_t)c.nt * step] = (hdC[j] + hdC[TILE + j]) + img[IC3::B3];
    } else {
      // critic (ppo.jl:128) as fp16x2 on two waves: wave 4 + mo owns hidden rows 32·mo … of both layers (the split3 kernel's actor structure)
      const int mo = wave - 4, j = lane & 31, hf = lane >> 5;
      const float* img = imgC0 + lds_off;
      const float4 cv = reinterpret_cast<const float4*>(xcur)[j];
      const float cx[4] = {cv.x, cv.y, cv.z, cv.w};
      f32x16 acc = load16(img + IC::B1C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < D / 2; ++ks) {
        const float bv = hf ? cx[2 * ks + 1] : cx[2 * ks];
        acc = mfma32(img[IC::WF1 + (mo * (D / 2) + ks) * 64 + lane], bv, acc);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float xb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xb[i] = tanh_exp2_arg(acc[8 * q + i], X2_ACT_SCALE);
        const P2 mine = split2(xb);
        f16x8* dst = pcsC + ((mo * 2 + q) * 2) * 64 + lane;
        dst[0] = mine.hi; dst[64] = mine.lo;
      }
      __syncthreads();                                                                        // (1)
      acc = load16(img + IC::B2C + hf * 32 + 16 * mo);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {     // every k-step's B pieces come back from LDS, this wave's own too: the kernel stays inside 128 registers (see the attribute above)
        const f16x8* src = pcsC + (((ks >> 1) * 2 + (ks & 1)) * 2) * 64 + lane;
        P2 bq; bq.hi = src[0]; bq.lo = src[64];
        acc = mfma_x2(load_wfrag2(img + IC::WF2H, mo, ks, lane), bq, acc);
      }
      const f32x4* w = reinterpret_cast<const f32x4*>(img + IC::W3 + hf * 32 + 16 * mo);
      float accv = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = w[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) accv = __builtin_fmaf(wv[i], tanh_exp2(acc[q * 4 + i], TWO_LOG2E * X2_FWD_UNSCALE, 1.0f), accv);
      }
      accv = add32(accv);
      if (hf == 0) hdC[mo * TILE + j] = accv;
      __syncthreads();                                                                        // (2)
      if (mo == 0 && hf == 0 && ok) a.value[(size_t)(blockIdx.x * TILE + j) + (size_t)c.nt * step] = (hdC[j] + hdC[TILE + j]) + img[IC::B3];
    }
    __syncthreads();                                                                          // (3) next observations are published, the exchange buffers are free
  }
  if (wave == 0) {
    if (writer) {
      reinterpret_cast<float4*>(a.env_state)[e] = make_float4(s[0], s[1], s[2], s[3]);
      reinterpret_cast<float4*>(a.cur_obs)[e] = make_float4(co[0], co[1], co[2], co[3]);
      a.env_t[e] = t_env; a.next_done[e] = nd; a.ep_return[e] = ep_ret; a.ep_length[e] = ep_len;
      // value[] was written by wave 4 of this block; the step loop's closing __syncthreads() made it visible here
      if (a.fuse_gae) gae_tail_compat(a, e);
    }
    st_n = wave_sum(st_n); st_ret = wave_sum(st_ret); st_len = wave_sum(st_len);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) st_max = fmax(st_max, __shfl_xor(st_max, o, 64));
    if (lane == 0 && st_n > 0.0) {
      atomicAdd(&a.ep_stats[0], st_n); atomicAdd(&a.ep_stats[1], st_ret); atomicAdd(&a.ep_stats[2], st_len);
      atomicMax(reinterpret_cast<unsigned long long*>(&a.ep_stats[3]), (unsigned long long)__double_as_longlong(st_max));
    }
  }
}

template <int D, int A>
static size_t act_smem() { return sizeof(float) * (NetImage<D, A, false>::SIZE + NetImage<D, 1, false>::SIZE); }

static int check_shape(crl_ppo* h) {
  if (h->cfg.obs_dim != 4 || h->cfg.n_act != 2 || h->cfg.hidden != 64) {
    set_error("this build of libcleanrl_hip supports obs_dim=4, n_act=2, hidden=64 (2x64 MLP) only");
    return 1;
  }
  return 0;
}

static int grid_for_tiles(int n, int wpb) {
  int tiles = (n + TILE - 1) / TILE;
  int blocks = (tiles + wpb - 1) / wpb;
  return blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
}

int launch_policy_act(crl_ppo* h, const float* obs_d, const double* u_d, int n, int32_t* action_d, float* logprob_d,
                      float* value_d) {
  if (h->wide) return wide_policy_act(h, obs_d, u_d, n, action_d, logprob_d, value_d);
  if (check_shape(h)) return 1;
  const int wpb = 4;
  const size_t smem = act_smem<4, 2>();
  hipLaunchKernelGGL((policy_act_kernel<4, 2>), dim3(grid_for_tiles(n, wpb)), dim3(64 * wpb), smem, h->stream,
                     h->params, obs_d, u_d, n, action_d, logprob_d, value_d);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_logprob_actions(crl_ppo* h, const float* obs_d, const int32_t* act_d, int n, float* logprob_d, float* ent_d) {
  if (h->wide) return wide_logprob_actions(h, obs_d, act_d, n, logprob_d, ent_d);
  if (check_shape(h)) return 1;
  const int wpb = 4;
  const size_t smem = sizeof(float) * NetImage<4, 2, false>::SIZE;
  hipLaunchKernelGGL((logprob_actions_kernel<4, 2>), dim3(grid_for_tiles(n, wpb)), dim3(64 * wpb), smem, h->stream, h->params, obs_d, act_d, n, logprob_d, ent_d);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_next_value(crl_ppo* h) {
  if (h->wide) return wide_next_value(h);
  if (check_shape(h)) return 1;
  const int wpb = 4;
  const size_t smem = sizeof(float) * NetImage<4, 1, false>::SIZE;
  hipLaunchKernelGGL((value_kernel<4>), dim3(grid_for_tiles(h->dc.nt, wpb)), dim3(64 * wpb), smem, h->stream, h->params + h->Pa, h->cur_obs, h->dc.nt,
                     h->next_value);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_env_reset(crl_ppo* h) {
  if (h->cfg.env_kind == CRL_ENV_SYNTHETIC) return wide_env_reset(h);
  if (h->cfg.env_kind != CRL_ENV_CARTPOLE) { set_error("crl_env_reset: envs are stepped by the caller (CRL_ENV_EXTERNAL)"); return 1; }
  hipLaunchKernelGGL(env_reset_kernel, dim3((h->dc.nt + 255) / 256), dim3(256), 0, h->stream, h->dc, h->env_state, h->env_t,
                     h->cur_obs, h->next_done, h->ep_return, h->ep_length, h->ep_stats);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_rollout(crl_ppo* h, bool fuse_gae) {
  if (h->cfg.env_kind == CRL_ENV_EXTERNAL) { set_error("crl_rollout_run: envs are stepped by the caller (CRL_ENV_EXTERNAL)"); return 1; }
  if (h->wide) return wide_rollout(h);
  if (check_shape(h)) return 1;
  CRL_HIP_CHECK(hipMemsetAsync(h->ep_stats, 0, 4 * sizeof(double), h->stream));
  RolloutArgs a;
  a.c = h->dc; a.params = h->params;
  a.obs = h->obs; a.action = h->action; a.logprob = h->logprob; a.reward = h->reward; a.terminal = h->terminal; a.value = h->value;
  a.env_state = h->env_state; a.env_t = h->env_t; a.cur_obs = h->cur_obs; a.next_done = h->next_done;
  a.ep_return = h->ep_return; a.ep_length = h->ep_length; a.ep_stats = h->ep_stats; a.iteration = (uint64_t)h->iteration;
  a.ring = h->ep_ring; a.ring_count = h->ep_ring_count; a.ring_cap = h->ep_ring_cap;
  a.adv = h->adv; a.ret = h->ret; a.fuse_gae = fuse_gae ? 1 : 0; a.gamma = h->cfg.gamma; a.gl = h->cfg.gamma * h->cfg.gae_lambda;
  if (h->ep_ring_cap > 0) CRL_HIP_CHECK(hipMemsetAsync(h->ep_ring_count, 0, sizeof(uint32_t), h->stream));
  // one wave per 32 envs; spread waves over all 256 CUs before stacking them inside a block
  const int tiles = (h->dc.nt + TILE - 1) / TILE;
  int wpb = tiles >= 2048 ? 8 : (tiles >= 1024 ? 4 : (tiles >= 512 ? 2 : 1));
  a.stagger = wpb == 8 ? (int)opt(h, OPT_ROLLOUT_STAGGER) : 0;
  const int blocks = (tiles + wpb - 1) / wpb;
  ProfScope ps(h, CRL_K_ROLLOUT);
  const int split = (int)opt(h, OPT_ROLLOUT_SPLIT);
  const bool small = tiles <= (int)opt(h, OPT_ROLLOUT_SPLIT_MAX_TILES);
  a.range_err = h->vfix + 5;
  // rollout_split = 4 (default): by size — six waves per tile wherever the split kernels apply (<= rollout_split_max_tiles = 512 tiles = 16384 envs): 2.6 against
  // 3.4 µs per step at 4096 / 8192 envs, 4.25 against 4.8 at 16384 (two six-wave blocks per CU: they fit because the kernel stays inside 128 registers);
  // profiles/r06_rollout_split6_ab.txt. 1 selects the three-wave kernel.
  const bool six = split == 3 || split == 4;
  if (gemm_x2(h) && six && small) {
    // six waves per tile: the actor's hidden rows over four waves (16x16x32 products), the critic's over two
    const size_t smem = sizeof(float) * (NetImageA16<2>::SIZE + NetImageX3<4, 1, false>::SIZE + 2 * TILE * 4 + 2 * 2 * 3 * 64 * 2 * 2 + 2 * 2 * 3 * 64 * 4 + 4 * 2 * TILE + 2 * TILE + 2 * 2 * TILE + 4 * 2 * TILE + 4);
    hipLaunchKernelGGL((rollout_split6_kernel<2>), dim3(tiles), dim3(384), smem, h->stream, a);
  } else if (gemm_x2(h) && (split == 1 || split == 4) && small) {
    // three waves per tile: the actor's hidden rows split over two waves, the critic (fp16x2) on the third
    const size_t smem = sizeof(float) * (NetImageX3<4, 2, false>::SIZE + NetImageX3<4, 1, false>::SIZE + 2 * TILE * 4 + 2 * 2 * 3 * 64 * 4 + 2 * 64 + 4);
    hipLaunchKernelGGL((rollout_split3_kernel<2>), dim3(tiles), dim3(192), smem, h->stream, a);
  } else if (split != 0 && small) {   // two waves per tile (actor + env | critic), all bf16x3: rollout_split = 2, or gemm = 1
    const size_t smem = sizeof(float) * (NetImageX3<4, 2, false>::SIZE + NetImageX3<4, 1, false>::SIZE + 2 * TILE * 4);
    hipLaunchKernelGGL((rollout_split_kernel<2>), dim3(tiles), dim3(128), smem, h->stream, a);
  } else {
    const size_t smem = sizeof(float) * (NetImageX3<4, 2, false>::SIZE + NetImageX3<4, 1, false>::SIZE + 4);   // 53 KB: critic image sized for its bf16x3 fallback
    if (gemm_x2(h)) hipLaunchKernelGGL((rollout_cartpole_kernel<2, true>), dim3(blocks), dim3(64 * wpb), smem, h->stream, a);
    else hipLaunchKernelGGL((rollout_cartpole_kernel<2, false>), dim3(blocks), dim3(64 * wpb), smem, h->stream, a);
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace crl
