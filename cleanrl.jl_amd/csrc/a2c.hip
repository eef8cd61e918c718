// a2c.hip — src/algorithms/a2c.jl on the GPU (C ABI: the crl_a2c_* block of include/cleanrl_hip.h).
//
// The reference steps ONE CartPoleEnv{Float64} and trains at an episode end once more than min_replay_size transitions
// have piled up (a2c.jl:53-111). Float64 observations through Float32-weight Dense layers promote every activation,
// loss and cotangent to Float64 — so this file is plain Float64 VALU code, one wave per sample (lane = hidden unit):
//   a2c_collect_kernel   the whole `for global_step` body between two updates in ONE launch: actor forward, softmax,
//                        Distributions-style categorical draw, env step, Buffer.add!, episode bookkeeping
//   a2c_forward_kernel   actor / critic forward of the training batch (hidden activations kept for the pullbacks)
//   a2c_loss_kernel      discounted_future_rewards (a2c.jl:13-24), advantage, both losses, output cotangents
//   a2c_backward_kernel  δ2, δ1 per sample;  a2c_wgrad_kernel  parameter gradients, summed over samples in sample order
//   clipnorm_adam_kernel (optim.hip) on the critic's six arrays, then on the actor's (a2c.jl:88,98)
// A training batch is ≤ 2·min_replay_size samples of a 2x64 MLP (≈0.1 GFLOP): there is nothing for the matrix pipes to
// win; what matters is one launch per phase and no per-step host round trip.
// Contraction is off so that dot products round like the CPU oracle's (mul, then add, in k order).
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "common.hpp"
#include "ppo_ctx.hpp"

struct crl_a2c;

namespace crl {
int launch_clipnorm_adam_range(hipStream_t st, float* params, const float* grads, float* m, float* v, double* betap,
                               const int* off13, int a0, int a1, double eta);

constexpr int AH = 64, AD = 4, AA = 2;       // hidden / obs / actions of the reference's CartPole networks
constexpr int A2C_MAX_EPS = 4096;

struct A2CCtl {
  double env[4]; int32_t env_t; int32_t size, ptr, pending;
  int64_t global_step; double episode_return; int64_t episode_length;
  int32_t n_eps, pad; int64_t taken;
  double final_value, critic_loss, actor_loss;
};

struct A2CDev {
  crl_a2c_config cfg; int cap;
  const float* params;
  A2CCtl* ctl; crl_a2c_episode* eps;
  double* rb_state; int32_t* rb_action; double* rb_reward; uint8_t* rb_terminal;
};

// ------------------------------------------------------------------------------------------------------
// Float64 pieces (oracle: a2c_tanh_fast, a2c_sin, a2c_cos, a2c_cartpole_step)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double tanh_fast64(double x) {
#pragma clang fp contract(off)
  const double exp2x = exp(x + x);
  const double y = (exp2x - 1.0) / (exp2x + 1.0);
  const double x2 = x * x;
  double p = -0.008697141630499953;
  p = p * x2 + 0.02186660872609521;
  p = p * x2 + -0.05396823125794372;
  p = p * x2 + 0.13333333325511604;
  p = p * x2 + -0.33333333333324583;
  p = p * x2 + 1.0;
  const double ypoly = x * p;
  if (x2 > 900.0) return (double)((x > 0.0) - (x < 0.0));
  return x2 < 0.017 ? ypoly : y;
}
__device__ __forceinline__ double sin64(double x) {
  const double c[10] = {-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                        -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, -1.0 / 121645100408832000.0,
                        1.0 / 51090942171709440000.0};
  const double x2 = x * x;
  double p = c[9];
#pragma unroll
  for (int i = 8; i >= 0; --i) p = __builtin_fma(p, x2, c[i]);
  return __builtin_fma(x * x2, p, x);
}
__device__ __forceinline__ double cos64(double x) {
  const double c[10] = {-0.5, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600.0,
                        -1.0 / 87178291200.0, 1.0 / 20922789888000.0, -1.0 / 6402373705728000.0,
                        1.0 / 2432902008176640000.0};
  const double x2 = x * x;
  double p = c[9];
#pragma unroll
  for (int i = 8; i >= 0; --i) p = __builtin_fma(p, x2, c[i]);
  return __builtin_fma(x2, p, 1.0);
}
__device__ __forceinline__ bool cartpole_step64(double (&s)[4], int& t, int action, int max_steps) {
#pragma clang fp contract(off)
  const double gravity = 9.8, masspole = 0.1, totalmass = 1.1, halflength = 0.5, pml = 0.05;
  const double forcemag = 10.0, dt = 0.02, ththr = 12.0 * 2.0 * 3.141592653589793 / 360.0, xthr = 2.4;
  t += 1;
  const double force = action == 1 ? forcemag : -forcemag;
  const double xdot = s[1], theta = s[2], thetadot = s[3];
  const double costheta = cos64(theta), sintheta = sin64(theta);
  const double tmp = (force + pml * thetadot * thetadot * sintheta) / totalmass;
  const double thetaacc = (gravity * sintheta - costheta * tmp) / (halflength * (4.0 / 3.0 - masspole * costheta * costheta / totalmass));
  const double xacc = tmp - pml * thetaacc * costheta / totalmass;
  s[0] += dt * xdot;
  s[1] += dt * xacc;
  s[2] += dt * thetadot;
  s[3] += dt * thetaacc;
  return (fabs(s[0]) > xthr) || (fabs(s[2]) > ththr) || (t > max_steps);
}
__device__ __forceinline__ void env_reset64(double (&s)[4], uint64_t seed, uint64_t gstep, uint32_t stream) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = 0.1 * u53(philox_env(seed, (uint32_t)i, gstep, stream)) - 0.05;
}

// Row `lane` of the first two layers of one network, in registers
struct NetRow { float w1[AD]; float b1; float w2[AH]; float b2; };
__device__ __forceinline__ void load_row(const float* __restrict__ P, int lane, NetRow& r) {
#pragma unroll
  for (int k = 0; k < AD; ++k) r.w1[k] = P[lane + AH * k];
  r.b1 = P[AH * AD + lane];
  const float* W2 = P + AH * AD + AH;
#pragma unroll
  for (int j = 0; j < AH; ++j) r.w2[j] = W2[lane + AH * j];
  r.b2 = W2[AH * AH + lane];
}
// Dense(W, b, tanh_fast) twice for ONE sample, lane = unit; hs is a 64-double LDS scratch (ends holding h2)
__device__ __forceinline__ void hidden_forward(const NetRow& r, const double (&x)[AD], double* hs, int lane, double& h1, double& h2) {
#pragma clang fp contract(off)
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < AD; ++k) acc += (double)r.w1[k] * x[k];
  acc += (double)r.b1;
  h1 = tanh_fast64(acc);
  __syncthreads();
  hs[lane] = h1;
  __syncthreads();
  acc = 0.0;
#pragma unroll
  for (int j = 0; j < AH; ++j) acc += (double)r.w2[j] * hs[j];
  acc += (double)r.b2;
  h2 = tanh_fast64(acc);
  __syncthreads();
  hs[lane] = h2;
  __syncthreads();
}
// head: out[a] = Σ_k W3[a, k]·h2[k] + b3[a] in k order, on lane a (a < n_out); results land in zs
__device__ __forceinline__ void head_forward(const float* __restrict__ W3, const float* __restrict__ b3, int n_out, const double* hs,
                                             double* zs, int lane) {
#pragma clang fp contract(off)
  if (lane < n_out) {
    double acc = 0.0;
#pragma unroll 16
    for (int k = 0; k < AH; ++k) acc += (double)W3[lane + n_out * k] * hs[k];   // W3 may live in LDS (collect kernel)
    zs[lane] = acc + (double)b3[lane];
  }
  __syncthreads();
}

constexpr int NET_SIZE_A = AH * AD + AH + AH * AH + AH + AA * AH + AA;   // actor parameter count
__device__ __forceinline__ const float* net_base(const float* params, int net) { return params + (net ? NET_SIZE_A : 0); }

// ------------------------------------------------------------------------------------------------------
// a2c.jl:53-74,104-110 — everything between two training updates, one launch, one wave
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) a2c_collect_kernel(A2CDev a, int64_t max_env_steps) {
#pragma clang fp contract(off)
  __shared__ double hs[AH];
  __shared__ double zs[AA];
  const int lane = threadIdx.x;
  const float* P = net_base(a.params, 0);
  NetRow r;
  load_row(P, lane, r);
  // the head's rows sit in LDS for the whole launch: its 64-term dot product would otherwise wait for a global load per term
  __shared__ float w3s[AA * AH + AA];
  {
    const float* W3g = P + AH * AD + AH + AH * AH + AH;
    for (int i = lane; i < AA * AH + AA; i += 64) w3s[i] = W3g[i];
  }
  __syncthreads();
  const float* W3 = w3s;
  const float* b3 = w3s + AA * AH;
  A2CCtl c = *a.ctl;
  c.n_eps = 0;
  int64_t taken = 0;
  while (taken < max_env_steps && c.global_step < a.cfg.total_timesteps && !c.pending) {
    c.global_step += 1;                                               // a2c.jl:53
    taken += 1;
    const uint64_t gstep = (uint64_t)c.global_step;
    double x[AD] = {c.env[0], c.env[1], c.env[2], c.env[3]};          // a2c.jl:55 deepcopy(state(env))
    double h1, h2;
    hidden_forward(r, x, hs, lane, h1, h2);
    head_forward(W3, b3, AA, hs, zs, lane);
    // softmax (a2c.jl:56) and rand(Categorical(probs)) (a2c.jl:57-58): cp = p[1]; while cp <= draw && i < n
    const double z0 = zs[0], z1 = zs[1];
    const double m = z1 > z0 ? z1 : z0;
    const double e0 = exp(z0 - m), e1 = exp(z1 - m);
    const double ssum = e0 + e1;
    const double p0 = e0 / ssum;
    const double draw = u53(philox_env(a.cfg.seed, 0u, gstep, 0u));
    const int action = (p0 <= draw) ? 1 : 0;
    const bool done = cartpole_step64(c.env, c.env_t, action, a.cfg.max_steps);   // a2c.jl:60
    const double rew = done ? 0.0 : 1.0;                              // reward(env) (0 on the terminating step)
    if (lane == 0) {                                                  // a2c.jl:62-68 Buffer.add!
#pragma unroll
      for (int k = 0; k < AD; ++k) a.rb_state[(size_t)AD * c.ptr + k] = x[k];
      a.rb_action[c.ptr] = action; a.rb_reward[c.ptr] = rew; a.rb_terminal[c.ptr] = done ? 1 : 0;
    }
    c.ptr = c.ptr + 1 >= a.cap ? 0 : c.ptr + 1;
    c.size = c.size + 1 > a.cap ? a.cap : c.size + 1;
    c.episode_return += rew; c.episode_length += 1;                  // a2c.jl:71-72
    if (done) {                                                       // a2c.jl:74
      if (c.size > a.cfg.min_replay_size) {
        c.pending = 1;                                                // a2c.jl:75: the update runs next, on the un-reset env
      } else {
        if (lane == 0 && c.n_eps < A2C_MAX_EPS) {                     // a2c.jl:105-106
          a.eps[c.n_eps].episode_return = c.episode_return; a.eps[c.n_eps].episode_length = c.episode_length;
          a.eps[c.n_eps].global_step = c.global_step;
        }
        c.n_eps += 1;
        c.episode_length = 0; c.episode_return = 0.0;                 // a2c.jl:108
        env_reset64(c.env, a.cfg.seed, gstep, 1); c.env_t = 0;        // a2c.jl:109
      }
    }
  }
  c.taken = taken;
  if (lane == 0) *a.ctl = c;
}

// after the update: the deferred a2c.jl:102-109 (clear!, episode record, reset!)
__global__ void a2c_finish_kernel(A2CDev a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  A2CCtl c = *a.ctl;
  c.size = 0; c.ptr = 0;                                              // a2c.jl:102 Buffer.clear!
  if (c.n_eps < A2C_MAX_EPS) {
    a.eps[c.n_eps].episode_return = c.episode_return; a.eps[c.n_eps].episode_length = c.episode_length;
    a.eps[c.n_eps].global_step = c.global_step;
  }
  c.n_eps += 1;
  c.episode_length = 0; c.episode_return = 0.0;
  env_reset64(c.env, a.cfg.seed, (uint64_t)c.global_step, 1); c.env_t = 0;
  c.pending = 0;
  *a.ctl = c;
}

// ------------------------------------------------------------------------------------------------------
// training batch: forward (one wave per sample)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) a2c_forward_kernel(const float* __restrict__ params, int net, const double* __restrict__ states,
                                                        int n, double* __restrict__ h1g, double* __restrict__ h2g, double* __restrict__ outg) {
  __shared__ double hs[AH];
  __shared__ double zs[AA];
  const int lane = threadIdx.x, b = blockIdx.x;
  if (b >= n) return;
  const float* P = net_base(params, net);
  const int n_out = net ? 1 : AA;
  NetRow r;
  load_row(P, lane, r);
  const float* W3 = P + AH * AD + AH + AH * AH + AH;
  const float* b3 = W3 + n_out * AH;
  const double x[AD] = {states[(size_t)AD * b], states[(size_t)AD * b + 1], states[(size_t)AD * b + 2], states[(size_t)AD * b + 3]};
  double h1, h2;
  hidden_forward(r, x, hs, lane, h1, h2);
  if (h1g) { h1g[(size_t)AH * b + lane] = h1; h2g[(size_t)AH * b + lane] = h2; }
  head_forward(W3, b3, n_out, hs, zs, lane);
  if (lane < n_out) outg[(size_t)n_out * b + lane] = zs[lane];
}

// ------------------------------------------------------------------------------------------------------
// discounted_future_rewards + losses + output cotangents (a2c.jl:13-24,79-97). One block; the reverse scan and the two
// loss sums run serially on thread 0 in the reference's order (n ≤ 2·min_replay_size), the rest per sample.
// ------------------------------------------------------------------------------------------------------
struct A2CLossArgs {
  const double* reward; const uint8_t* terminal; const int32_t* action; int n; double gamma;
  const double* v; const double* z; A2CCtl* ctl;
  double* G; double* adv; double* dvc; double* dza; double* lp_adv;
};
__device__ void returns_scan(const double* reward, const uint8_t* terminal, int n, double final_value, double gamma, double* G) {
#pragma clang fp contract(off)
  double next = terminal[n - 1] ? 0.0 : reward[n - 1] + gamma * final_value;
  G[n - 1] = next;
  for (int j = n - 2; j >= 0; --j) {
    next = terminal[j] ? 0.0 : reward[j] + gamma * next;
    G[j] = next;
  }
}
__global__ void __launch_bounds__(1024) a2c_loss_kernel(A2CLossArgs a) {
#pragma clang fp contract(off)
  const int n = a.n;
  if (threadIdx.x == 0) returns_scan(a.reward, a.terminal, n, a.ctl->final_value, a.gamma, a.G);
  __syncthreads();
  for (int b = threadIdx.x; b < n; b += blockDim.x) {
    const double adv = a.G[b] - a.v[b];                               // a2c.jl:85
    a.adv[b] = adv;
    a.dvc[b] = -2.0 * adv / (double)n;
    const double z0 = a.z[2 * (size_t)b], z1 = a.z[2 * (size_t)b + 1];
    const double m = z1 > z0 ? z1 : z0;
    const double e0 = exp(z0 - m), e1 = exp(z1 - m);
    const double s = e0 + e1;
    const double p0 = e0 / s, p1 = e1 / s;
    const int act = a.action[b];
    const double lp = log(act ? p1 : p0);                             // logpdf(Categorical(p), a) a2c.jl:95
    a.lp_adv[b] = lp * adv;
    const double k = -adv / (double)n;
    a.dza[2 * (size_t)b] = k * ((act == 0 ? 1.0 : 0.0) - p0);
    a.dza[2 * (size_t)b + 1] = k * ((act == 1 ? 1.0 : 0.0) - p1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double cl = 0.0, al = 0.0;
    for (int b = 0; b < n; ++b) { cl += a.adv[b] * a.adv[b]; al += a.lp_adv[b]; }
    a.ctl->critic_loss = cl / (double)n;                              // a2c.jl:86
    a.ctl->actor_loss = -(al / (double)n);                            // a2c.jl:96
  }
}
// standalone discounted_future_rewards (crl_a2c_discounted_future_rewards)
__global__ void a2c_returns_kernel(const double* reward, const uint8_t* terminal, int n, double final_value, double gamma, double* G) {
  if (threadIdx.x == 0 && blockIdx.x == 0) returns_scan(reward, terminal, n, final_value, gamma, G);
}

// δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²), δ1 = (W2ᵀ·δ2) ⊙ (1 − h1²): one wave per sample, lane = unit
__global__ void __launch_bounds__(64) a2c_backward_kernel(const float* __restrict__ params, int net, const double* __restrict__ dout, int n,
                                                         const double* __restrict__ h1g, const double* __restrict__ h2g,
                                                         double* __restrict__ d2g, double* __restrict__ d1g) {
#pragma clang fp contract(off)
  __shared__ double ds[AH];
  const int k = threadIdx.x, b = blockIdx.x;
  if (b >= n) return;
  const float* P = net_base(params, net);
  const int n_out = net ? 1 : AA;
  const float* W2 = P + AH * AD + AH;
  const float* W3 = W2 + AH * AH + AH;
  double s = 0.0;
  for (int q = 0; q < n_out; ++q) s += (double)W3[q + n_out * k] * dout[(size_t)n_out * b + q];
  const double h2 = h2g[(size_t)AH * b + k];
  const double d2 = s * (1.0 - h2 * h2);
  d2g[(size_t)AH * b + k] = d2;
  ds[k] = d2;
  __syncthreads();
  s = 0.0;
#pragma unroll 8
  for (int i = 0; i < AH; ++i) s += (double)W2[i + AH * k] * ds[i];
  const double h1 = h1g[(size_t)AH * b + k];
  d1g[(size_t)AH * b + k] = s * (1.0 - h1 * h1);
}

// parameter gradients of one network: thread = one parameter, samples summed in sample order, Float32 projection
__global__ void __launch_bounds__(256) a2c_wgrad_kernel(int net, const double* __restrict__ states, const double* __restrict__ dout, int n,
                                                       const double* __restrict__ h1g, const double* __restrict__ h2g,
                                                       const double* __restrict__ d2g, const double* __restrict__ d1g,
                                                       float* __restrict__ grads) {
#pragma clang fp contract(off)
  const int n_out = net ? 1 : AA;
  const int ob1 = AH * AD, oW2 = ob1 + AH, ob2 = oW2 + AH * AH, oW3 = ob2 + AH, ob3 = oW3 + n_out * AH, total = ob3 + n_out;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  double g = 0.0;
  if (idx < ob1) { const int i = idx % AH, k = idx / AH; for (int b = 0; b < n; ++b) g += d1g[(size_t)AH * b + i] * states[(size_t)AD * b + k]; }
  else if (idx < oW2) { const int i = idx - ob1; for (int b = 0; b < n; ++b) g += d1g[(size_t)AH * b + i]; }
  else if (idx < ob2) { const int q = idx - oW2, i = q % AH, k = q / AH; for (int b = 0; b < n; ++b) g += d2g[(size_t)AH * b + i] * h1g[(size_t)AH * b + k]; }
  else if (idx < oW3) { const int i = idx - ob2; for (int b = 0; b < n; ++b) g += d2g[(size_t)AH * b + i]; }
  else if (idx < ob3) { const int q = idx - oW3, aa = q % n_out, k = q / n_out; for (int b = 0; b < n; ++b) g += dout[(size_t)n_out * b + aa] * h2g[(size_t)AH * b + k]; }
  else { const int aa = idx - ob3; for (int b = 0; b < n; ++b) g += dout[(size_t)n_out * b + aa]; }
  grads[(net ? NET_SIZE_A : 0) + idx] = (float)g;
}

__global__ void a2c_init_kernel(A2CDev a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  A2CCtl c;
  memset(&c, 0, sizeof(c));
  env_reset64(c.env, a.cfg.seed, 0, 2);                               // a2c.jl:52 reset!(env)
  *a.ctl = c;
}

}  // namespace crl

// ------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------
struct crl_a2c {
  crl_a2c_config cfg;
  int device = 0; int cap = 0; int64_t P = 0;
  bool params_set = false;   // a2c.jl:37 has happened: parameters were uploaded or crl_a2c_init_params ran (a fresh handle holds zeros)
  hipStream_t stream = nullptr;
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr; double* betap = nullptr;
  crl::A2CCtl* ctl = nullptr; crl_a2c_episode* eps = nullptr;
  double* rb_state = nullptr; int32_t* rb_action = nullptr; double* rb_reward = nullptr; uint8_t* rb_terminal = nullptr;
  // training workspace, [·, cap]
  double *h1[2] = {nullptr, nullptr}, *h2[2] = {nullptr, nullptr}, *d1 = nullptr, *d2 = nullptr;
  double *vout = nullptr, *z = nullptr, *G = nullptr, *adv = nullptr, *dvc = nullptr, *dza = nullptr, *lp_adv = nullptr;
  int off[13];
};

namespace crl {
static A2CDev dev_args(crl_a2c* h) {
  A2CDev a;
  a.cfg = h->cfg; a.cap = h->cap; a.params = h->params; a.ctl = h->ctl; a.eps = h->eps;
  a.rb_state = h->rb_state; a.rb_action = h->rb_action; a.rb_reward = h->rb_reward; a.rb_terminal = h->rb_terminal;
  return a;
}
template <typename T>
static int aalloc(T** p, size_t n) {
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  CRL_HIP_CHECK(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}
#define A2C_GUARD(h)                                           \
  if (!(h)) { crl::set_error("null crl_a2c handle"); return 1; } \
  CRL_HIP_CHECK(hipSetDevice((h)->device));

static int a2c_update(crl_a2c* h, int n) {
  A2CDev a = dev_args(h);
  hipStream_t st = h->stream;
  // final_value = critic(state(env))[1] on the un-reset terminal state (a2c.jl:78)
  hipLaunchKernelGGL(a2c_forward_kernel, dim3(1), dim3(64), 0, st, h->params, 1, h->ctl->env, 1, (double*)nullptr, (double*)nullptr,
                     &h->ctl->final_value);
  hipLaunchKernelGGL(a2c_forward_kernel, dim3(n), dim3(64), 0, st, h->params, 1, h->rb_state, n, h->h1[1], h->h2[1], h->vout);
  hipLaunchKernelGGL(a2c_forward_kernel, dim3(n), dim3(64), 0, st, h->params, 0, h->rb_state, n, h->h1[0], h->h2[0], h->z);
  A2CLossArgs l;
  l.reward = h->rb_reward; l.terminal = h->rb_terminal; l.action = h->rb_action; l.n = n; l.gamma = h->cfg.gamma;
  l.v = h->vout; l.z = h->z; l.ctl = h->ctl; l.G = h->G; l.adv = h->adv; l.dvc = h->dvc; l.dza = h->dza; l.lp_adv = h->lp_adv;
  hipLaunchKernelGGL(a2c_loss_kernel, dim3(1), dim3(1024), 0, st, l);
  CRL_HIP_CHECK(hipGetLastError());
  for (int net = 1; net >= 0; --net) {   // critic update first (a2c.jl:81-88), then the actor's (a2c.jl:90-98)
    const double* dout = net ? h->dvc : h->dza;
    const int n_out = net ? 1 : AA, total = AH * AD + AH + AH * AH + AH + n_out * AH + n_out;
    hipLaunchKernelGGL(a2c_backward_kernel, dim3(n), dim3(64), 0, st, h->params, net, dout, n, h->h1[net], h->h2[net], h->d2, h->d1);
    hipLaunchKernelGGL(a2c_wgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, st, net, h->rb_state, dout, n, h->h1[net], h->h2[net],
                       h->d2, h->d1, h->grads);
    CRL_HIP_CHECK(hipGetLastError());
  }
  if (launch_clipnorm_adam_range(st, h->params, h->grads, h->m, h->v, h->betap, h->off, 6, 12, h->cfg.lr)) return 1;
  if (launch_clipnorm_adam_range(st, h->params, h->grads, h->m, h->v, h->betap, h->off, 0, 6, h->cfg.lr)) return 1;
  hipLaunchKernelGGL(a2c_finish_kernel, dim3(1), dim3(1), 0, st, a);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}
}  // namespace crl

using namespace crl;

extern "C" {

int32_t crl_a2c_create(const crl_a2c_config* cfg, int32_t device, crl_a2c** out) {
  if (!cfg || !out) { set_error("crl_a2c_create: null argument"); return 1; }
  *out = nullptr;
  if (cfg->min_replay_size < 1 || cfg->max_steps < 1 || cfg->total_timesteps < 0) { set_error("crl_a2c_create: bad sizes"); return 1; }
  if (cfg->min_replay_size < cfg->max_steps + 1) {
    set_error("crl_a2c_create: min_replay_size must be >= max_steps + 1 (the 2x replay buffer must hold min_replay_size + one whole episode)");
    return 1;
  }
  int ndev = 0;
  CRL_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { set_error("crl_a2c_create: no such HIP device (no GPU → no CPU fallback)"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  crl_a2c* h = new (std::nothrow) crl_a2c();
  if (!h) { set_error("out of host memory"); return 1; }
  h->cfg = *cfg; h->device = device; h->cap = 2 * cfg->min_replay_size;       // a2c.jl:46
  const int sizes[12] = {AH * AD, AH, AH * AH, AH, AA * AH, AA, AH * AD, AH, AH * AH, AH, AH, 1};
  h->off[0] = 0;
  for (int i = 0; i < 12; ++i) h->off[i + 1] = h->off[i] + sizes[i];
  h->P = h->off[12];
  hipError_t se = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (se != hipSuccess) { set_error(std::string("hipStreamCreate: ") + hipGetErrorString(se)); delete h; return 1; }
  const size_t cap = (size_t)h->cap, P = (size_t)h->P;
  int rc = 0;
  rc |= aalloc(&h->params, P); rc |= aalloc(&h->grads, P); rc |= aalloc(&h->m, P); rc |= aalloc(&h->v, P); rc |= aalloc(&h->betap, 24);
  rc |= aalloc(&h->ctl, 1); rc |= aalloc(&h->eps, A2C_MAX_EPS);
  rc |= aalloc(&h->rb_state, cap * AD); rc |= aalloc(&h->rb_action, cap); rc |= aalloc(&h->rb_reward, cap); rc |= aalloc(&h->rb_terminal, cap);
  for (int n = 0; n < 2; ++n) { rc |= aalloc(&h->h1[n], cap * AH); rc |= aalloc(&h->h2[n], cap * AH); }
  rc |= aalloc(&h->d1, cap * AH); rc |= aalloc(&h->d2, cap * AH);
  rc |= aalloc(&h->vout, cap); rc |= aalloc(&h->z, cap * AA); rc |= aalloc(&h->G, cap); rc |= aalloc(&h->adv, cap);
  rc |= aalloc(&h->dvc, cap); rc |= aalloc(&h->dza, cap * AA); rc |= aalloc(&h->lp_adv, cap);
  if (rc) { crl_a2c_destroy(h); return 1; }
  double bp[24];
  for (int i = 0; i < 12; ++i) { bp[2 * i] = 0.9; bp[2 * i + 1] = 0.999; }
  CRL_HIP_CHECK(hipMemcpy(h->betap, bp, sizeof(bp), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(a2c_init_kernel, dim3(1), dim3(1), 0, h->stream, dev_args(h));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  *out = h;
  return 0;
}

int32_t crl_a2c_destroy(crl_a2c* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->params, h->grads, h->m, h->v, h->betap, h->ctl, h->eps, h->rb_state, h->rb_action, h->rb_reward, h->rb_terminal,
                  h->h1[0], h->h1[1], h->h2[0], h->h2[1], h->d1, h->d2, h->vout, h->z, h->G, h->adv, h->dvc, h->dza, h->lp_adv};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int32_t crl_a2c_param_count(const crl_a2c* h, int64_t* n) {
  if (!h || !n) { set_error("null argument"); return 1; }
  *n = h->P;
  return 0;
}
int32_t crl_a2c_write_params(crl_a2c* h, const float* params, size_t n) {
  A2C_GUARD(h);
  if (!params || n != (size_t)h->P) { set_error("crl_a2c_write_params: expected " + std::to_string(h->P) + " floats"); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(h->params, params, n * 4, hipMemcpyHostToDevice, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  h->params_set = true;
  return 0;
}

int32_t crl_a2c_init_params(crl_a2c* h, uint64_t seed) {
  A2C_GUARD(h);
  std::vector<float> w((size_t)h->P);
  if (crl_make_actor_critic(AD, AA, AH, seed, w.data(), w.size())) return 1;
  return crl_a2c_write_params(h, w.data(), w.size());
}
int32_t crl_a2c_read_params(crl_a2c* h, float* params, size_t n) {
  A2C_GUARD(h);
  if (!params || n != (size_t)h->P) { set_error("crl_a2c_read_params: expected " + std::to_string(h->P) + " floats"); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(params, h->params, n * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}
static int read_ctl(crl_a2c* h, A2CCtl* c) {
  CRL_HIP_CHECK(hipMemcpyAsync(c, h->ctl, sizeof(A2CCtl), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}
int32_t crl_a2c_read_env(crl_a2c* h, double* state4, int64_t* global_step, int32_t* rb_size) {
  A2C_GUARD(h);
  A2CCtl c;
  if (read_ctl(h, &c)) return 1;
  if (state4) for (int i = 0; i < 4; ++i) state4[i] = c.env[i];
  if (global_step) *global_step = c.global_step;
  if (rb_size) *rb_size = c.size;
  return 0;
}
int32_t crl_a2c_read_buffer(crl_a2c* h, double* state, int32_t* action, double* reward, uint8_t* terminal, int32_t capacity) {
  A2C_GUARD(h);
  A2CCtl c;
  if (read_ctl(h, &c)) return 1;
  if (capacity < c.size) { set_error("crl_a2c_read_buffer: caller arrays hold fewer than rb.size transitions"); return 1; }
  const size_t n = (size_t)c.size;
  if (n == 0) return 0;
  if (!state || !action || !reward || !terminal) { set_error("crl_a2c_read_buffer: null argument"); return 1; }
  CRL_HIP_CHECK(hipMemcpyAsync(state, h->rb_state, n * AD * 8, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(action, h->rb_action, n * 4, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(reward, h->rb_reward, n * 8, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipMemcpyAsync(terminal, h->rb_terminal, n, hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  return 0;
}

int32_t crl_a2c_run_until_update(crl_a2c* h, int64_t max_env_steps, crl_a2c_train_stats* stats, crl_a2c_episode* eps,
                                 int32_t max_eps, int32_t* n_eps, int64_t* steps_taken) {
  A2C_GUARD(h);
  if (!h->params_set) { set_error("crl_a2c_run_until_update: parameters not set — crl_a2c_write_params or crl_a2c_init_params first (a2c.jl:37; a fresh handle holds zeros)"); return 1; }
  if (!stats || !n_eps || (max_eps > 0 && !eps) || max_eps < 0) { set_error("crl_a2c_run_until_update: bad arguments"); return 1; }
  stats->actor_loss = 0.0; stats->critic_loss = 0.0; stats->n = 0; stats->trained = 0;
  *n_eps = 0;
  if (steps_taken) *steps_taken = 0;
  if (max_env_steps <= 0) return 0;
  hipLaunchKernelGGL(a2c_collect_kernel, dim3(1), dim3(64), 0, h->stream, dev_args(h), max_env_steps);
  CRL_HIP_CHECK(hipGetLastError());
  A2CCtl c;
  if (read_ctl(h, &c)) return 1;
  if (steps_taken) *steps_taken = c.taken;
  if (c.pending) {
    const int n = c.size;
    if (a2c_update(h, n)) return 1;
    if (read_ctl(h, &c)) return 1;
    stats->actor_loss = c.actor_loss; stats->critic_loss = c.critic_loss; stats->n = n; stats->trained = 1;
  }
  int ne = c.n_eps < A2C_MAX_EPS ? c.n_eps : A2C_MAX_EPS;
  if (ne > max_eps) ne = max_eps;
  if (ne > 0) {
    CRL_HIP_CHECK(hipMemcpyAsync(eps, h->eps, sizeof(crl_a2c_episode) * (size_t)ne, hipMemcpyDeviceToHost, h->stream));
    CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  }
  *n_eps = ne;
  return 0;
}

int32_t crl_a2c_discounted_future_rewards(int32_t device, const double* rewards, const uint8_t* terminals, int32_t n,
                                          double final_value, double gamma, double* out) {
  if (n < 0) { set_error("crl_a2c_discounted_future_rewards: negative size"); return 1; }
  if (n == 0) return 0;
  if (!rewards || !terminals || !out) { set_error("crl_a2c_discounted_future_rewards: null argument"); return 1; }
  CRL_HIP_CHECK(hipSetDevice(device));
  char* buf = nullptr;
  const size_t N = (size_t)n, o_r = 0, o_g = N * 8, o_t = 2 * N * 8, total = o_t + N;
  CRL_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&buf), total));
  int rc = 0;
  hipError_t e;
  if ((e = hipMemcpy(buf + o_r, rewards, N * 8, hipMemcpyHostToDevice)) != hipSuccess) rc = 1;
  if (!rc && (e = hipMemcpy(buf + o_t, terminals, N, hipMemcpyHostToDevice)) != hipSuccess) rc = 1;
  if (!rc) {
    hipLaunchKernelGGL(a2c_returns_kernel, dim3(1), dim3(1), 0, nullptr, (const double*)(buf + o_r), (const uint8_t*)(buf + o_t), n,
                       final_value, gamma, (double*)(buf + o_g));
    if ((e = hipDeviceSynchronize()) != hipSuccess) rc = 1;
  }
  if (!rc && (e = hipMemcpy(out, buf + o_g, N * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = 1;
  if (rc) set_error(std::string("crl_a2c_discounted_future_rewards: ") + hipGetErrorString(e));
  (void)hipFree(buf);
  return rc;
}

}  // extern "C"
