// dqn.hip — src/algorithms/dqn.jl on the GPU (C ABI: the crl_dqn_* block of include/cleanrl_hip.h).
//
// The reference interleaves ONE CartPoleEnv{Float64} step with a 120-sample update every train_freq steps. The
// `for global_step` body (dqn.jl:57-119) runs here as a fixed sequence of launches per train_freq steps, enqueued without
// any host read-back in between: dqn_collect_kernel (ε-greedy action — q_net forward only on greedy steps —, env step,
// ring-buffer add, episode bookkeeping, up to the next training step), then the minibatch draw without replacement, both
// forwards, TD target, Flux.mse, the pullbacks, Adam and the hard target copy as multi-CU kernels.
// Float64 arithmetic with Float32 weights like the reference (see a2c.hip); relu has no transcendental, every sum runs
// in the oracle's order with contraction off ⇒ the run is BIT-IDENTICAL to oracle/dqn_oracle.c.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "common.hpp"
#include "ppo_ctx.hpp"

struct crl_dqn;

namespace crl {

constexpr int QH1 = 120, QH2 = 84, QD = 4, QA = 2;
constexpr int QoW1 = 0, Qob1 = QH1 * QD, QoW2 = Qob1 + QH1, Qob2 = QoW2 + QH2 * QH1, QoW3 = Qob2 + QH2, Qob3 = QoW3 + QA * QH2;
constexpr int QP = Qob3 + QA;
static_assert(QP == CRL_DQN_PARAM_COUNT, "parameter count");
constexpr int DQN_MAX_EPS = 8192, DQN_MAX_LOSSES = 4096, DQN_MAX_BATCH = 1024, DQN_MAX_CAP = 1 << 16;

struct DQNCtl {
  double env[4]; double episode_return, last_loss;
  int64_t global_step, episode_length, n_updates, taken, ptr, size, budget;
  int32_t env_t, n_eps, n_losses, train_pending;
};

struct DQNDev {
  crl_dqn_config cfg;
  float *q, *t, *grads, *m, *v; double* betap;
  DQNCtl* ctl; crl_dqn_episode* eps; crl_dqn_loss_record* losses;
  double *rb_state, *rb_next, *rb_reward; int32_t* rb_action; uint8_t* rb_terminal;
  double *H1, *H2, *Q, *dz, *sq, *d2, *d1;   // [2][120·k], [2][84·k], [2][2·k], [k], [k], [84·k], [120·k]
  int32_t* idx;                               // [k] minibatch indices into the ring
};

// CartPoleEnv{Float64}: shared with a2c.hip's restatement (oracle: a2c_cartpole_step)
__device__ __forceinline__ double dq_sin64(double x) {
  const double c[10] = {-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                        -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, -1.0 / 121645100408832000.0,
                        1.0 / 51090942171709440000.0};
  const double x2 = x * x;
  double p = c[9];
#pragma unroll
  for (int i = 8; i >= 0; --i) p = __builtin_fma(p, x2, c[i]);
  return __builtin_fma(x * x2, p, x);
}
__device__ __forceinline__ double dq_cos64(double x) {
  const double c[10] = {-0.5, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600.0,
                        -1.0 / 87178291200.0, 1.0 / 20922789888000.0, -1.0 / 6402373705728000.0,
                        1.0 / 2432902008176640000.0};
  const double x2 = x * x;
  double p = c[9];
#pragma unroll
  for (int i = 8; i >= 0; --i) p = __builtin_fma(p, x2, c[i]);
  return __builtin_fma(x2, p, 1.0);
}
__device__ __forceinline__ bool dq_cartpole_step(double* s, int& t, int action, int max_steps) {
#pragma clang fp contract(off)
  const double gravity = 9.8, masspole = 0.1, totalmass = 1.1, halflength = 0.5, pml = 0.05;
  const double forcemag = 10.0, dt = 0.02, ththr = 12.0 * 2.0 * 3.141592653589793 / 360.0, xthr = 2.4;
  t += 1;
  const double force = action == 1 ? forcemag : -forcemag;
  const double xdot = s[1], theta = s[2], thetadot = s[3];
  const double costheta = dq_cos64(theta), sintheta = dq_sin64(theta);
  const double tmp = (force + pml * thetadot * thetadot * sintheta) / totalmass;
  const double thetaacc = (gravity * sintheta - costheta * tmp) / (halflength * (4.0 / 3.0 - masspole * costheta * costheta / totalmass));
  const double xacc = tmp - pml * thetaacc * costheta / totalmass;
  s[0] += dt * xdot;
  s[1] += dt * xacc;
  s[2] += dt * thetadot;
  s[3] += dt * thetaacc;
  return (fabs(s[0]) > xthr) || (fabs(s[2]) > ththr) || (t > max_steps);
}
__device__ __forceinline__ void dq_env_reset(double* s, uint64_t seed, uint64_t gstep, uint32_t stream) {
#pragma clang fp contract(off)
  for (int i = 0; i < 4; ++i) s[i] = 0.1 * u53(philox_env(seed, (uint32_t)i, gstep, stream)) - 0.05;
}
__device__ __forceinline__ double dq_linear_schedule(double start_e, double end_e, double duration, double t) {
#pragma clang fp contract(off)
  const double slope = (end_e - start_e) / duration;
  const double v = slope * t + start_e;
  return v > end_e ? v : end_e;
}

// ------------------------------------------------------------------------------------------------------
// One cycle = [collect: env steps up to and including the next training step] → [update phases]. The schedule is
// deterministic (dqn.jl:93), every decision is taken on the device (kernels early-exit on the control block), so the
// host simply enqueues ⌈steps / train_freq⌉ + 1 identical cycles without ever reading anything back in between.
// The update phases are ordinary multi-CU launches: an update is ≈5 M Float64 multiply-adds, ≈0.5 ms on ONE CU but a
// few µs per phase on the whole chip. Per-element arithmetic and summation order are exactly the oracle's.
// ------------------------------------------------------------------------------------------------------
__global__ void dqn_begin_kernel(DQNDev a, int64_t max_env_steps) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  a.ctl->n_eps = 0; a.ctl->n_losses = 0; a.ctl->taken = 0; a.ctl->budget = max_env_steps;
}

// dqn.jl:57-93: env steps until a training step is due (train_pending = 1), the budget or total_timesteps is used up
__global__ void __launch_bounds__(128) dqn_collect_kernel(DQNDev a) {
#pragma clang fp contract(off)
  __shared__ DQNCtl c;
  __shared__ double obs[QD], h1s[QH1], h2s[QH2], qs[QA];
  __shared__ int go, need_q, action_s;
  __shared__ double eps_s;
  __shared__ float wq[QP];   // q_net for the greedy steps of this launch (its dot products read one weight per term)
  const int tid = threadIdx.x;
  if (tid == 0) c = *a.ctl;
  __syncthreads();
  // nothing to do (budget used up / training still pending): skip the 44 KB weight load
  if (c.train_pending || c.taken >= c.budget || c.global_step >= a.cfg.total_timesteps) return;
  for (int p = tid; p < QP; p += 128) wq[p] = a.q[p];
  __syncthreads();
  while (true) {
    if (tid == 0) {
      go = (!c.train_pending && c.taken < c.budget && c.global_step < a.cfg.total_timesteps) ? 1 : 0;
      if (go) {
        c.global_step += 1; c.taken += 1;                                            // dqn.jl:57
        const uint64_t gstep = (uint64_t)c.global_step;
        for (int i = 0; i < QD; ++i) obs[i] = c.env[i];                              // dqn.jl:58
        eps_s = dq_linear_schedule(a.cfg.epsilon_start, a.cfg.epsilon_end, a.cfg.epsilon_duration, (double)c.global_step);
        if (u53(philox_env(a.cfg.seed, 0u, gstep, 0u)) < eps_s) {                    // dqn.jl:61-62
          need_q = 0;
          action_s = (int)(philox_env(a.cfg.seed, 0u, gstep, 4u).x >> 31);
        } else need_q = 1;
      }
    }
    __syncthreads();
    if (!go) break;
    if (need_q) {                                                                    // dqn.jl:64 qs = q_net(obs)
      if (tid < QH1) {
        double acc = 0.0;
        for (int kk = 0; kk < QD; ++kk) acc += (double)wq[QoW1 + tid + QH1 * kk] * obs[kk];
        acc += (double)wq[Qob1 + tid];
        h1s[tid] = acc > 0.0 ? acc : 0.0;
      }
      __syncthreads();
      if (tid < QH2) {
        double acc = 0.0;
#pragma unroll 8
        for (int kk = 0; kk < QH1; ++kk) acc += (double)wq[QoW2 + tid + QH2 * kk] * h1s[kk];
        acc += (double)wq[Qob2 + tid];
        h2s[tid] = acc > 0.0 ? acc : 0.0;
      }
      __syncthreads();
      if (tid < QA) {
        double acc = 0.0;
#pragma unroll 4
        for (int kk = 0; kk < QH2; ++kk) acc += (double)wq[QoW3 + tid + QA * kk] * h2s[kk];
        qs[tid] = acc + (double)wq[Qob3 + tid];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const uint64_t gstep = (uint64_t)c.global_step;
      const int action = need_q ? (qs[1] > qs[0] ? 1 : 0) : action_s;               // dqn.jl:65 argmax: first maximum
      const bool done = dq_cartpole_step(c.env, c.env_t, action, a.cfg.max_steps);  // dqn.jl:68
      const double rew = done ? 0.0 : 1.0;
      const size_t p = (size_t)c.ptr;                                                // dqn.jl:71-78 Buffer.add!
      for (int i = 0; i < QD; ++i) { a.rb_state[QD * p + i] = obs[i]; a.rb_next[QD * p + i] = c.env[i]; }
      a.rb_action[p] = action; a.rb_reward[p] = rew; a.rb_terminal[p] = done ? 1 : 0;
      c.ptr = c.ptr + 1 >= a.cfg.buffer_size ? 0 : c.ptr + 1;
      c.size = c.size + 1 > a.cfg.buffer_size ? a.cfg.buffer_size : c.size + 1;
      c.episode_return += rew; c.episode_length += 1;                               // dqn.jl:81-82
      if (done) {                                                                    // dqn.jl:83-90
        if (c.n_eps < DQN_MAX_EPS) {
          a.eps[c.n_eps].episode_return = c.episode_return; a.eps[c.n_eps].episode_length = c.episode_length;
          a.eps[c.n_eps].global_step = c.global_step; a.eps[c.n_eps].epsilon = eps_s;
        }
        c.n_eps += 1;
        c.episode_length = 0; c.episode_return = 0.0;
        dq_env_reset(c.env, a.cfg.seed, gstep, 1); c.env_t = 0;
      }
      c.train_pending = (c.global_step > a.cfg.min_buff_size && c.global_step % a.cfg.train_freq == 0) ? 1 : 0;   // dqn.jl:93
    }
    __syncthreads();
  }
  if (tid == 0) *a.ctl = c;
}

// Buffer.sample(rb, batch_size) (dqn.jl:94): self-avoiding draws, candidates in counter order (oracle: dqn_sample_indices)
__global__ void __launch_bounds__(1024) dqn_sample_kernel(DQNDev a) {
  if (!a.ctl->train_pending) return;
  __shared__ uint32_t bitmap[DQN_MAX_CAP / 32];
  __shared__ int cand[1024];
  __shared__ int got, cbase;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int k = (int)a.cfg.batch_size, n = (int)a.ctl->size;
  const uint64_t gstep = (uint64_t)a.ctl->global_step;
  for (int w = tid; w < (n + 31) / 32; w += nth) bitmap[w] = 0u;
  if (tid == 0) { got = 0; cbase = 0; }
  __syncthreads();
  while (got < k) {
    {
      const uint32_t cc = (uint32_t)(cbase + tid);
      const u32x4 o = philox(cc, (uint32_t)gstep, (uint32_t)(gstep >> 32), 0xD9u, (uint32_t)a.cfg.seed, (uint32_t)(a.cfg.seed >> 32));
      cand[tid] = (int)(((uint64_t)o.x * (uint64_t)n) >> 32);
    }
    __syncthreads();
    if (tid == 0) {
      int g = got;
      for (int i = 0; i < nth && g < k; ++i) {
        const int j = cand[i];
        if (!((bitmap[j >> 5] >> (j & 31)) & 1u)) { bitmap[j >> 5] |= 1u << (j & 31); a.idx[g++] = j; }
      }
      got = g; cbase += nth;
    }
    __syncthreads();
  }
}

// forward of target_net(next_state) (net 1) and q_net(state) (net 0), one output element per thread (dqn.jl:99,104)
__global__ void __launch_bounds__(256) dqn_fwd1_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= 2 * QH1 * k) return;
  const int net = o / (QH1 * k), r = o - net * (QH1 * k), b = r / QH1, i = r - b * QH1;
  const float* p = net ? a.t : a.q;
  const double* x = (net ? a.rb_next : a.rb_state) + (size_t)QD * a.idx[b];
  double acc = 0.0;
  for (int kk = 0; kk < QD; ++kk) acc += (double)p[QoW1 + i + QH1 * kk] * x[kk];
  acc += (double)p[Qob1 + i];
  a.H1[(size_t)net * QH1 * k + r] = acc > 0.0 ? acc : 0.0;
}
__global__ void __launch_bounds__(256) dqn_fwd2_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= 2 * QH2 * k) return;
  const int net = o / (QH2 * k), r = o - net * (QH2 * k), b = r / QH2, j = r - b * QH2;
  const float* p = net ? a.t : a.q;
  const double* h = a.H1 + (size_t)net * QH1 * k + (size_t)QH1 * b;
  double acc = 0.0;
#pragma unroll 8
  for (int kk = 0; kk < QH1; ++kk) acc += (double)p[QoW2 + j + QH2 * kk] * h[kk];
  acc += (double)p[Qob2 + j];
  a.H2[(size_t)net * QH2 * k + r] = acc > 0.0 ? acc : 0.0;
}
__global__ void __launch_bounds__(256) dqn_fwd3_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= 2 * QA * k) return;
  const int net = o / (QA * k), r = o - net * (QA * k), b = r / QA, aa = r - b * QA;
  const float* p = net ? a.t : a.q;
  const double* h = a.H2 + (size_t)net * QH2 * k + (size_t)QH2 * b;
  double acc = 0.0;
#pragma unroll 4
  for (int kk = 0; kk < QH2; ++kk) acc += (double)p[QoW3 + aa + QA * kk] * h[kk];
  a.Q[(size_t)net * QA * k + r] = acc + (double)p[Qob3 + aa];
}
// TD target, mse, output cotangent (dqn.jl:99-107); one block, the loss is summed in sample order by thread 0
__global__ void __launch_bounds__(1024) dqn_td_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size;
  for (int b = threadIdx.x; b < k; b += blockDim.x) {
    const double tq0 = a.Q[(size_t)QA * k + QA * b], tq1 = a.Q[(size_t)QA * k + QA * b + 1];
    const double next_q = tq1 > tq0 ? tq1 : tq0;
    const int s = a.idx[b];
    const double td = a.rb_reward[s] + a.cfg.gamma * next_q * (1.0 - (double)a.rb_terminal[s]);
    const double diff = td - a.Q[QA * b + a.rb_action[s]];
    a.sq[b] = diff * diff;
    a.dz[b] = -2.0 * diff / (double)k;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double loss = 0.0;
    for (int b = 0; b < k; ++b) loss += a.sq[b];
    a.ctl->last_loss = loss / (double)k;
  }
}
__global__ void __launch_bounds__(256) dqn_bwd2_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= QH2 * k) return;
  const int b = o / QH2, j = o - b * QH2;
  const double s = (double)a.q[QoW3 + a.rb_action[a.idx[b]] + QA * j] * a.dz[b];
  a.d2[o] = a.H2[o] > 0.0 ? s : 0.0;
}
__global__ void __launch_bounds__(256) dqn_bwd1_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= QH1 * k) return;
  const int b = o / QH1, kk = o - b * QH1;
  const double* dd = a.d2 + (size_t)QH2 * b;
  double s = 0.0;
#pragma unroll 6
  for (int j = 0; j < QH2; ++j) s += (double)a.q[QoW2 + j + QH2 * kk] * dd[j];
  a.d1[o] = a.H1[o] > 0.0 ? s : 0.0;
}
// one parameter per thread, samples summed in sample order, Float32 projection
__global__ void __launch_bounds__(64) dqn_wgrad_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int k = (int)a.cfg.batch_size, p = blockIdx.x * 64 + threadIdx.x;
  if (p >= QP) return;
  // every loop is unrolled 30x: the operands were written by the previous launches, so they come from the Infinity Cache
  // at ≈2 µs per dependent round trip — 4 rounds of 60 loads in flight instead of 15 of 16. The Float64 adds stay in
  // sample order. Adding an exact 0.0 for samples of the other action leaves the sum unchanged.
  double g = 0.0;
  if (p < Qob1) {
    const int i = p % QH1, kk = p / QH1;
#pragma unroll 30
    for (int b = 0; b < k; ++b) g += a.d1[(size_t)QH1 * b + i] * a.rb_state[(size_t)QD * a.idx[b] + kk];
  } else if (p < QoW2) {
    const int i = p - Qob1;
#pragma unroll 30
    for (int b = 0; b < k; ++b) g += a.d1[(size_t)QH1 * b + i];
  } else if (p < Qob2) {
    const int r = p - QoW2, j = r % QH2, kk = r / QH2;
#pragma unroll 30
    for (int b = 0; b < k; ++b) g += a.d2[(size_t)QH2 * b + j] * a.H1[(size_t)QH1 * b + kk];
  } else if (p < QoW3) {
    const int j = p - Qob2;
#pragma unroll 30
    for (int b = 0; b < k; ++b) g += a.d2[(size_t)QH2 * b + j];
  } else if (p < Qob3) {
    const int r = p - QoW3, aa = r % QA, j = r / QA;
#pragma unroll 30
    for (int b = 0; b < k; ++b) {
      const double t = a.dz[b] * a.H2[(size_t)QH2 * b + j];
      if (a.rb_action[a.idx[b]] == aa) g += t;
    }
  } else {
    const int aa = p - Qob3;
#pragma unroll 30
    for (int b = 0; b < k; ++b) {
      const double t = a.dz[b];
      if (a.rb_action[a.idx[b]] == aa) g += t;
    }
  }
  a.grads[p] = (float)g;
}
// Flux Adam(lr) (dqn.jl:41,109) + hard target copy (dqn.jl:111-113) + loss record (dqn.jl:115-117); one block
__global__ void __launch_bounds__(1024) dqn_adam_kernel(DQNDev a) {
#pragma clang fp contract(off)
  if (!a.ctl->train_pending) return;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int64_t gstep = a.ctl->global_step;
  const bool copy = gstep % a.cfg.target_net_freq == 0;
  for (int p = tid; p < QP; p += nth) {
    const int arr = p < Qob1 ? 0 : p < QoW2 ? 1 : p < Qob2 ? 2 : p < QoW3 ? 3 : p < Qob3 ? 4 : 5;
    const double b1 = 0.9, b2 = 0.999, epsn = 1e-8;
    const double bp0 = a.betap[2 * arr], bp1 = a.betap[2 * arr + 1];
    const double gg = (double)a.grads[p];
    const float mi = (float)(b1 * (double)a.m[p] + (1 - b1) * gg);
    const float vi = (float)(b2 * (double)a.v[p] + (1 - b2) * gg * gg);
    a.m[p] = mi; a.v[p] = vi;
    const double delta = (double)mi / (1 - bp0) / (sqrt((double)vi / (1 - bp1)) + epsn) * a.cfg.lr;
    const float w = a.q[p] - (float)delta;
    a.q[p] = w;
    if (copy) a.t[p] = w;
  }
  __syncthreads();
  if (tid < 12) a.betap[tid] = a.betap[tid] * ((tid & 1) ? 0.999 : 0.9);
  if (tid == 0) {
    DQNCtl* c = a.ctl;
    c->n_updates += 1;
    if (gstep % a.cfg.log_frequency == 0) {
      if (c->n_losses < DQN_MAX_LOSSES) { a.losses[c->n_losses].global_step = gstep; a.losses[c->n_losses].loss = c->last_loss; }
      c->n_losses += 1;
    }
    c->train_pending = 0;
  }
}

__global__ void dqn_init_kernel(DQNDev a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  DQNCtl c;
  memset(&c, 0, sizeof(c));
  dq_env_reset(c.env, a.cfg.seed, 0, 2);                                             // dqn.jl:56 reset!(env)
  *a.ctl = c;
}

// q_net(obs) on caller-supplied observations: one thread per (observation, layer unit) would be overkill — a wave per
// observation, lanes striding the units
__global__ void __launch_bounds__(128) dqn_q_kernel(const float* __restrict__ q, const double* __restrict__ obs, int n, double* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double h1s[QH1], h2s[QH2];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= n) return;
  if (tid < QH1) {
    double acc = 0.0;
    for (int kk = 0; kk < QD; ++kk) acc += (double)q[QoW1 + tid + QH1 * kk] * obs[(size_t)QD * b + kk];
    acc += (double)q[Qob1 + tid];
    h1s[tid] = acc > 0.0 ? acc : 0.0;
  }
  __syncthreads();
  if (tid < QH2) {
    double acc = 0.0;
    for (int kk = 0; kk < QH1; ++kk) acc += (double)q[QoW2 + tid + QH2 * kk] * h1s[kk];
    acc += (double)q[Qob2 + tid];
    h2s[tid] = acc > 0.0 ? acc : 0.0;
  }
  __syncthreads();
  if (tid < QA) {
    double acc = 0.0;
    for (int kk = 0; kk < QH2; ++kk) acc += (double)q[QoW3 + tid + QA * kk] * h2s[kk];
    out[(size_t)QA * b + tid] = acc + (double)q[Qob3 + tid];
  }
}

}  // namespace crl

struct crl_dqn {
  crl_dqn_config cfg;
  int device = 0;
  bool params_set = false;   // dqn.jl:39-40 has happened: q_net was uploaded or crl_dqn_init_params ran (a fresh handle holds zeros)
  hipStream_t stream = nullptr;
  crl::DQNDev d;
  void* stage = nullptr; size_t stage_bytes = 0;
  hipGraphExec_t cycle_exec = nullptr; bool graph_failed = false;   // CRL_DQN_GRAPH=1 replays the cycle as a hipGraph
};

using namespace crl;

#define DQN_GUARD(h)                                          \
  if (!(h)) { set_error("null crl_dqn handle"); return 1; }   \
  CRL_HIP_CHECK(hipSetDevice((h)->device));

template <typename T>
static int qalloc(T** p, size_t n) {
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  CRL_HIP_CHECK(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

static void enqueue_cycle(hipStream_t st, const DQNDev& d, int k) {
  hipLaunchKernelGGL(dqn_collect_kernel, dim3(1), dim3(128), 0, st, d);
  hipLaunchKernelGGL(dqn_sample_kernel, dim3(1), dim3(1024), 0, st, d);
  hipLaunchKernelGGL(dqn_fwd1_kernel, dim3((2 * QH1 * k + 255) / 256), dim3(256), 0, st, d);
  hipLaunchKernelGGL(dqn_fwd2_kernel, dim3((2 * QH2 * k + 255) / 256), dim3(256), 0, st, d);
  hipLaunchKernelGGL(dqn_fwd3_kernel, dim3((2 * QA * k + 255) / 256), dim3(256), 0, st, d);
  hipLaunchKernelGGL(dqn_td_kernel, dim3(1), dim3(1024), 0, st, d);
  hipLaunchKernelGGL(dqn_bwd2_kernel, dim3((QH2 * k + 255) / 256), dim3(256), 0, st, d);
  hipLaunchKernelGGL(dqn_bwd1_kernel, dim3((QH1 * k + 255) / 256), dim3(256), 0, st, d);
  hipLaunchKernelGGL(dqn_wgrad_kernel, dim3((QP + 63) / 64), dim3(64), 0, st, d);
  hipLaunchKernelGGL(dqn_adam_kernel, dim3(1), dim3(1024), 0, st, d);
}

extern "C" {

int32_t crl_dqn_create(const crl_dqn_config* cfg, int32_t device, crl_dqn** out) {
  if (!cfg || !out) { set_error("crl_dqn_create: null argument"); return 1; }
  *out = nullptr;
  if (cfg->batch_size < 1 || cfg->batch_size > DQN_MAX_BATCH) { set_error("crl_dqn_create: batch_size must be in 1..1024"); return 1; }
  if (cfg->buffer_size < cfg->batch_size || cfg->buffer_size > DQN_MAX_CAP) { set_error("crl_dqn_create: buffer_size must be in batch_size..65536"); return 1; }
  if (cfg->min_buff_size < cfg->batch_size) {
    set_error("crl_dqn_create: min_buff_size must be >= batch_size (Buffer.sample asserts n <= rb.size, replay_buffer.jl:41)"); return 1;
  }
  if (cfg->train_freq < 1 || cfg->target_net_freq < 1 || cfg->log_frequency < 1 || cfg->max_steps < 1 || cfg->total_timesteps < 0 ||
      !(cfg->epsilon_duration > 0.0)) { set_error("crl_dqn_create: bad frequencies / sizes"); return 1; }
  int ndev = 0;
  CRL_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { set_error("crl_dqn_create: no such HIP device (no GPU → no CPU fallback)"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  crl_dqn* h = new (std::nothrow) crl_dqn();
  if (!h) { set_error("out of host memory"); return 1; }
  h->cfg = *cfg; h->device = device;
  { const char* e = getenv("CRL_DQN_GRAPH"); h->graph_failed = !(e && atoi(e) != 0); }   // hipGraph replay is opt-in: measured no faster
  memset(&h->d, 0, sizeof(h->d));
  h->d.cfg = *cfg;
  hipError_t se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { set_error(std::string("hipStreamCreate: ") + hipGetErrorString(se)); delete h; return 1; }
  const size_t cap = (size_t)cfg->buffer_size, k = (size_t)cfg->batch_size;
  DQNDev& d = h->d;
  int rc = 0;
  rc |= qalloc(&d.q, QP); rc |= qalloc(&d.t, QP); rc |= qalloc(&d.grads, QP); rc |= qalloc(&d.m, QP); rc |= qalloc(&d.v, QP);
  rc |= qalloc(&d.betap, 12); rc |= qalloc(&d.ctl, 1); rc |= qalloc(&d.eps, DQN_MAX_EPS); rc |= qalloc(&d.losses, DQN_MAX_LOSSES);
  rc |= qalloc(&d.rb_state, cap * QD); rc |= qalloc(&d.rb_next, cap * QD); rc |= qalloc(&d.rb_reward, cap);
  rc |= qalloc(&d.rb_action, cap); rc |= qalloc(&d.rb_terminal, cap);
  rc |= qalloc(&d.H1, 2 * QH1 * k); rc |= qalloc(&d.H2, 2 * QH2 * k); rc |= qalloc(&d.Q, 2 * QA * k);
  rc |= qalloc(&d.dz, k); rc |= qalloc(&d.sq, k); rc |= qalloc(&d.d2, QH2 * k); rc |= qalloc(&d.d1, QH1 * k); rc |= qalloc(&d.idx, k);
  if (rc) { crl_dqn_destroy(h); return 1; }
  double bp[12];
  for (int i = 0; i < 6; ++i) { bp[2 * i] = 0.9; bp[2 * i + 1] = 0.999; }
  CRL_HIP_CHECK(hipMemcpy(d.betap, bp, sizeof(bp), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dqn_init_kernel, dim3(1), dim3(1), 0, h->stream, d);
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  *out = h;
  return 0;
}

int32_t crl_dqn_destroy(crl_dqn* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->cycle_exec) (void)hipGraphExecDestroy(h->cycle_exec);
  DQNDev& d = h->d;
  void* ptrs[] = {d.q, d.t, d.grads, d.m, d.v, d.betap, d.ctl, d.eps, d.losses, d.rb_state, d.rb_next, d.rb_reward, d.rb_action,
                  d.rb_terminal, d.H1, d.H2, d.Q, d.dz, d.sq, d.d2, d.d1, d.idx, h->stage};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int32_t crl_dqn_write_params(crl_dqn* h, const float* q_params, size_t n) {
  DQN_GUARD(h);
  if (!q_params || n != (size_t)QP) { set_error("crl_dqn_write_params: expected 10934 floats"); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(h->d.q, q_params, n * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(h->d.t, q_params, n * 4, hipMemcpyHostToDevice, h->stream));   // dqn.jl:40 deepcopy(q_net)
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  h->params_set = true;
  return 0;
}
int32_t crl_dqn_init_params(crl_dqn* h, uint64_t seed) {
  DQN_GUARD(h);
  std::vector<float> w((size_t)QP);
  if (crl_dqn_make_nn(seed, w.data(), w.size())) return 1;
  return crl_dqn_write_params(h, w.data(), w.size());
}

int32_t crl_dqn_read_params(crl_dqn* h, float* q_params, float* target_params, size_t n) {
  DQN_GUARD(h);
  if (!q_params || n != (size_t)QP) { set_error("crl_dqn_read_params: expected 10934 floats"); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(q_params, h->d.q, n * 4, hipMemcpyDeviceToHost, h->stream));
  if (target_params) CRL_HIP_CHECK(hipMemcpyAsync(target_params, h->d.t, n * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}
int32_t crl_dqn_status_read(crl_dqn* h, crl_dqn_status* out) {
  DQN_GUARD(h);
  if (!out) { set_error("null argument"); return 1; }
  DQNCtl c;
  CRL_HIP_CHECK(hipMemcpyAsync(&c, h->d.ctl, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 4; ++i) out->env_state[i] = c.env[i];
  out->global_step = c.global_step; out->rb_size = c.size; out->n_updates = c.n_updates; out->last_loss = c.last_loss;
  return 0;
}

int32_t crl_dqn_run(crl_dqn* h, int64_t max_env_steps, crl_dqn_episode* eps, int32_t max_eps, int32_t* n_eps,
                    crl_dqn_loss_record* losses, int32_t max_losses, int32_t* n_losses, int64_t* steps_taken) {
  DQN_GUARD(h);
  if (!h->params_set) { set_error("crl_dqn_run: parameters not set — crl_dqn_write_params or crl_dqn_init_params first (dqn.jl:39; a fresh handle holds zeros)"); return 1; }
  if (!n_eps || !n_losses || max_eps < 0 || max_losses < 0 || (max_eps > 0 && !eps) || (max_losses > 0 && !losses)) {
    set_error("crl_dqn_run: bad arguments"); return 1;
  }
  *n_eps = 0; *n_losses = 0;
  if (steps_taken) *steps_taken = 0;
  if (max_env_steps <= 0) return 0;
  {
    const DQNDev& d = h->d;
    hipStream_t st = h->stream;
    const int k = (int)h->cfg.batch_size;
    // a cycle ends at the next multiple of train_freq; one spare cycle covers a call that starts mid-cycle
    const int64_t limit = h->cfg.total_timesteps < max_env_steps ? h->cfg.total_timesteps : max_env_steps;
    const int64_t cycles = limit / h->cfg.train_freq + 2;
    hipLaunchKernelGGL(dqn_begin_kernel, dim3(1), dim3(1), 0, st, d, max_env_steps);
    if (!h->cycle_exec && !h->graph_failed) {
      // one cycle = ten dependent launches with constant arguments: captured once, replayed as a hipGraph (the cycle is
      // bound by the dependent kernels themselves — 52.7 K vs 53.1 K steps/s without the graph — so this stays opt-in)
      hipGraph_t g = nullptr;
      bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        enqueue_cycle(st, d, k);
        ok = hipStreamEndCapture(st, &g) == hipSuccess && g != nullptr;
        if (ok) ok = hipGraphInstantiate(&h->cycle_exec, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
      }
      if (!ok) { h->cycle_exec = nullptr; h->graph_failed = true; (void)hipGetLastError(); }
    }
    for (int64_t cy = 0; cy < cycles; ++cy) {
      if (h->cycle_exec) CRL_HIP_CHECK(hipGraphLaunch(h->cycle_exec, st));
      else enqueue_cycle(st, d, k);
      if ((cy & 4095) == 4095) CRL_HIP_CHECK(hipStreamSynchronize(st));   // bound the queue depth of very long calls
    }
    CRL_HIP_CHECK(hipGetLastError());
  }
  DQNCtl c;
  CRL_HIP_CHECK(hipMemcpyAsync(&c, h->d.ctl, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (steps_taken) *steps_taken = c.taken;
  int ne = c.n_eps < DQN_MAX_EPS ? c.n_eps : DQN_MAX_EPS; if (ne > max_eps) ne = max_eps;
  int nl = c.n_losses < DQN_MAX_LOSSES ? c.n_losses : DQN_MAX_LOSSES; if (nl > max_losses) nl = max_losses;
  if (ne > 0) CRL_HIP_CHECK(hipMemcpyAsync(eps, h->d.eps, sizeof(crl_dqn_episode) * (size_t)ne, hipMemcpyDeviceToHost, h->stream));
  if (nl > 0) CRL_HIP_CHECK(hipMemcpyAsync(losses, h->d.losses, sizeof(crl_dqn_loss_record) * (size_t)nl, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  *n_eps = ne; *n_losses = nl;
  return 0;
}

int32_t crl_dqn_q_values(crl_dqn* h, const double* obs, int32_t n, double* q) {
  DQN_GUARD(h);
  if (!h->params_set) { set_error("crl_dqn_q_values: parameters not set — crl_dqn_write_params or crl_dqn_init_params first (dqn.jl:39)"); return 1; }
  if (n < 0 || (n > 0 && (!obs || !q))) { set_error("crl_dqn_q_values: bad arguments"); return 1; }
  if (n == 0) return 0;
  const size_t N = (size_t)n, need = N * (QD + QA) * 8;
  if (h->stage_bytes < need) {
    if (h->stage) CRL_HIP_CHECK(hipFree(h->stage));
    h->stage = nullptr; h->stage_bytes = 0;
    CRL_HIP_CHECK(hipMalloc(&h->stage, need));
    h->stage_bytes = need;
  }
  double* so = static_cast<double*>(h->stage);
  double* sq = so + N * QD;
  CRL_HIP_CHECK(hipMemcpyAsync(so, obs, N * QD * 8, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(dqn_q_kernel, dim3(n), dim3(128), 0, h->stream, h->d.q, so, n, sq);
  CRL_HIP_CHECK(hipGetLastError());
  CRL_HIP_CHECK(hipMemcpyAsync(q, sq, N * QA * 8, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"
