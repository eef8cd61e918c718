/*
 * a2c_oracle.c — CPU restatement of src/algorithms/a2c.jl (TEST INFRASTRUCTURE ONLY). See a2c_oracle.h.
 * Third-party semantics restated from the pinned packages are marked [3P-memory] (VERIFY_WITH_JULIA.md).
 */
#include "a2c_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ppo_oracle.h" /* orc_philox, orc_u53: the build's counter RNG (same streams as the PPO path) */

/* ------------------------------------------------------------------------------------------------ */
/* Float64 pieces                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
/* NNlib 0.8.21 tanh_fast(x::Float64) [3P-memory]: (exp(2x)-1)/(exp(2x)+1), a degree-5 polynomial in x² near zero
 * (x² < 0.017), sign(y) beyond x² > 900. */
double a2c_tanh_fast(double x) {
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
  if (x2 > 900.0) return (x > 0.0) - (x < 0.0);   /* sign(y); y = ±1 exactly there (exp(2x) overflows only for |x| > 354) */
  return x2 < 0.017 ? ypoly : y;
}

/* sin / cos of the pole angle: explicit Taylor polynomials (|x| ≤ 0.5: truncation < 1e-22), Horner with fma, so the
 * env is bit-identical on CPU and GPU (same device as the Float32 env of the PPO path). */
double a2c_sin(double x) {
  static const double c[10] = {-1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                               -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, -1.0 / 121645100408832000.0,
                               1.0 / 51090942171709440000.0};
  const double x2 = x * x;
  double p = c[9];
  for (int i = 8; i >= 0; --i) p = fma(p, x2, c[i]);
  return fma(x * x2, p, x);
}
double a2c_cos(double x) {
  static const double c[10] = {-0.5, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600.0,
                               -1.0 / 87178291200.0, 1.0 / 20922789888000.0, -1.0 / 6402373705728000.0,
                               1.0 / 2432902008176640000.0};
  const double x2 = x * x;
  double p = c[9];
  for (int i = 8; i >= 0; --i) p = fma(p, x2, c[i]);
  return fma(x2, p, 1.0);
}

/* CartPoleEnv{Float64} step — RLEnvs 0.6.12 [3P-memory]; same Euler update as orc_cartpole_step, all Float64.
 * action is 0-based here (Julia a == 2 ↔ 1). */
void a2c_cartpole_step(double* s, int32_t* t, int32_t action, int32_t max_steps, int32_t* done) {
  const double gravity = 9.8, masspole = 0.1, totalmass = 1.1, halflength = 0.5, pml = 0.05;
  const double forcemag = 10.0, dt = 0.02, ththr = 12.0 * 2.0 * 3.141592653589793 / 360.0, xthr = 2.4;
  *t += 1;
  const double force = action == 1 ? forcemag : -forcemag;
  const double xdot = s[1], theta = s[2], thetadot = s[3];
  const double costheta = a2c_cos(theta), sintheta = a2c_sin(theta);
  const double tmp = (force + pml * thetadot * thetadot * sintheta) / totalmass;
  const double thetaacc = (gravity * sintheta - costheta * tmp) / (halflength * (4.0 / 3.0 - masspole * costheta * costheta / totalmass));
  const double xacc = tmp - pml * thetaacc * costheta / totalmass;
  s[0] += dt * xdot;
  s[1] += dt * xacc;
  s[2] += dt * thetadot;
  s[3] += dt * thetaacc;
  *done = (fabs(s[0]) > xthr) || (fabs(s[2]) > ththr) || (*t > max_steps);
}

/* reset!: state = 0.1 * rand(rng, Float64, 4) .- 0.05 [3P-memory]; each draw 53 bits * 2^-53 */
static void env_reset(const a2c_config* c, double* s, uint64_t gstep, uint32_t stream) {
  for (int i = 0; i < 4; ++i) s[i] = 0.1 * orc_u53(c->seed, (uint32_t)i, gstep, stream) - 0.05;
}

static void offsets(const a2c_config* c, int32_t* o) {
  const int h = c->hidden, d = c->obs_dim, A = c->n_act;
  const int sizes[12] = {h * d, h, h * h, h, A * h, A, h * d, h, h * h, h, h, 1};
  o[0] = 0;
  for (int i = 0; i < 12; ++i) o[i + 1] = o[i] + sizes[i];
}

/* Dense(W::Matrix{Float32}, b::Vector{Float32}, σ)(x::Vector{Float64}) = σ.(W*x .+ b): Float64 accumulation in k order */
static void dense64(const float* W, const float* b, const double* x, int out, int in, double* y, int act) {
  for (int o = 0; o < out; ++o) {
    double acc = 0.0;
    for (int i = 0; i < in; ++i) acc += (double)W[o + out * i] * x[i];
    acc += (double)b[o];
    y[o] = act ? a2c_tanh_fast(acc) : acc;
  }
}
static void forward_h(const a2c_config* c, const float* params, int net, const double* x, double* out, double* h1, double* h2) {
  int32_t o[13];
  offsets(c, o);
  const int h = c->hidden, base = net ? 6 : 0, n_out = net ? 1 : c->n_act;
  dense64(params + o[base + 0], params + o[base + 1], x, h, c->obs_dim, h1, 1);
  dense64(params + o[base + 2], params + o[base + 3], h1, h, h, h2, 1);
  dense64(params + o[base + 4], params + o[base + 5], h2, n_out, h, out, 0);
}
void a2c_forward(const a2c_config* c, const float* params, int net, const double* x, double* out) {
  double h1[1024], h2[1024];
  forward_h(c, params, net, x, out, h1, h2);
}

/* NNlib softmax over a Float64 vector [3P-memory] */
static void softmax64(const double* z, int A, double* p) {
  double m = z[0];
  for (int a = 1; a < A; ++a) m = z[a] > m ? z[a] : m;
  double s = 0.0;
  for (int a = 0; a < A; ++a) { p[a] = exp(z[a] - m); s += p[a]; }
  for (int a = 0; a < A; ++a) p[a] = p[a] / s;
}

/* Distributions.jl rand(::Categorical) [3P-memory]: draw = rand(); cp = p[1]; i = 1; while cp <= draw && i < n; cp += p[i += 1] */
static int sample_categorical(const double* p, int A, double draw) {
  double cp = p[0];
  int i = 0;
  while (cp <= draw && i < A - 1) { i += 1; cp += p[i]; }
  return i;
}

/* a2c.jl:13-24 */
void a2c_discounted_future_rewards(const double* rewards, const uint8_t* terminals, int32_t n, double final_value,
                                   double gamma, double* out) {
  if (n <= 0) return;
  /* future_rewards[1] = last(terminals) ? 0.0 : last(rewards) + γ * final_value ; then walk the reversed rest */
  double next = terminals[n - 1] ? 0.0 : rewards[n - 1] + gamma * final_value;
  out[n - 1] = next;
  for (int j = n - 2; j >= 0; --j) {
    next = terminals[j] ? 0.0 : rewards[j] + gamma * next;
    out[j] = next;
  }
}

/* a2c.jl:81-97 + Zygote pullbacks. grads: Float32 projection of the Float64 cotangents, flat Flux order. */
void a2c_loss_grads(const a2c_config* c, const float* params, const double* states, const int32_t* actions,
                    const double* returns, int32_t n, float* grads, double* critic_loss, double* actor_loss, double* advantage) {
  int32_t o[13];
  offsets(c, o);
  const int h = c->hidden, d = c->obs_dim, A = c->n_act, P = o[12];
  double* g = (double*)calloc((size_t)P, sizeof(double));
  double* h1 = (double*)malloc(sizeof(double) * h);
  double* h2 = (double*)malloc(sizeof(double) * h);
  double* d2 = (double*)malloc(sizeof(double) * h);
  double* d1 = (double*)malloc(sizeof(double) * h);
  double closs = 0.0, aloss = 0.0;
  for (int net = 1; net >= 0; --net) {   /* critic first (a2c.jl:81-88), then actor with the captured advantage */
    const int base = net ? 6 : 0, n_out = net ? 1 : A;
    const float* W2 = params + o[base + 2];
    const float* W3 = params + o[base + 4];
    for (int b = 0; b < n; ++b) {
      const double* x = states + (size_t)d * b;
      double out[16], dout[16];
      forward_h(c, params, net, x, out, h1, h2);
      if (net) {
        const double adv = returns[b] - out[0];            /* a2c.jl:85 */
        advantage[b] = adv;
        closs += adv * adv;                                /* a2c.jl:86 mean(advantage .^ 2) */
        dout[0] = -2.0 * adv / (double)n;
      } else {
        double p[16];
        softmax64(out, A, p);
        const double lp = log(p[actions[b]]);              /* logpdf(Categorical(p), a) a2c.jl:95 */
        aloss += lp * advantage[b];
        const double k = -advantage[b] / (double)n;        /* d(-mean(lp .* adv))/d lp_b */
        for (int a = 0; a < A; ++a) dout[a] = k * ((a == actions[b] ? 1.0 : 0.0) - p[a]);
      }
      /* head */
      for (int a = 0; a < n_out; ++a) {
        g[o[base + 5] + a] += dout[a];
        for (int k = 0; k < h; ++k) g[o[base + 4] + a + n_out * k] += dout[a] * h2[k];
      }
      for (int k = 0; k < h; ++k) {
        double s = 0.0;
        for (int a = 0; a < n_out; ++a) s += (double)W3[a + n_out * k] * dout[a];
        d2[k] = s * (1.0 - h2[k] * h2[k]);
      }
      for (int i = 0; i < h; ++i) {
        g[o[base + 3] + i] += d2[i];
        for (int k = 0; k < h; ++k) g[o[base + 2] + i + h * k] += d2[i] * h1[k];
      }
      for (int k = 0; k < h; ++k) {
        double s = 0.0;
        for (int i = 0; i < h; ++i) s += (double)W2[i + h * k] * d2[i];
        d1[k] = s * (1.0 - h1[k] * h1[k]);
      }
      for (int i = 0; i < h; ++i) {
        g[o[base + 1] + i] += d1[i];
        for (int k = 0; k < d; ++k) g[o[base + 0] + i + h * k] += d1[i] * x[k];
      }
    }
  }
  for (int i = 0; i < P; ++i) grads[i] = (float)g[i];
  *critic_loss = closs / (double)n;
  *actor_loss = -(aloss / (double)n);
  free(g); free(h1); free(h2); free(d2); free(d1);
}

/* Flux 0.13.4 Optimiser(ClipNorm(0.5), Adam(η)) over arrays [a0, a1) (a2c.jl:38,88,98): same arithmetic as orc_clipnorm_adam */
static void clipnorm_adam_range(const int32_t* o, int a0, int a1, float* params, float* grads, float* m, float* v, double* betap,
                                double eta) {
  const double b1 = 0.9, b2 = 0.999, epsn = 1e-8, thresh = 0.5;
  for (int a = a0; a < a1; ++a) {
    double ss = 0.0;
    for (int i = o[a]; i < o[a + 1]; ++i) ss += (double)grads[i] * grads[i];
    const float nrm = (float)sqrt(ss);
    if ((double)nrm > thresh) {
      const double sc = thresh / (double)nrm;
      for (int i = o[a]; i < o[a + 1]; ++i) grads[i] = (float)((double)grads[i] * sc);
    }
    double* bp = betap + 2 * a;
    for (int i = o[a]; i < o[a + 1]; ++i) {
      const double gg = grads[i];
      m[i] = (float)(b1 * (double)m[i] + (1 - b1) * gg);
      v[i] = (float)(b2 * (double)v[i] + (1 - b2) * gg * gg);
      const double delta = (double)m[i] / (1 - bp[0]) / (sqrt((double)v[i] / (1 - bp[1])) + epsn) * eta;
      params[i] = params[i] - (float)delta;
    }
    bp[0] *= b1; bp[1] *= b2;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Loop state (a2c.jl:32-52)                                                                          */
/* ------------------------------------------------------------------------------------------------ */
struct a2c_state {
  a2c_config c;
  int32_t off[13];
  float *params, *grads, *m, *v;
  double betap[24];
  double env[4]; int32_t env_t;
  /* replay buffer (replay_buffer.jl:15-37) */
  int32_t cap, ptr, size;
  double* rb_state; int32_t* rb_action; double* rb_reward; uint8_t* rb_terminal;
  double episode_return; int64_t episode_length, global_step;
};

a2c_state* a2c_create(const a2c_config* c) {
  a2c_state* s = (a2c_state*)calloc(1, sizeof(a2c_state));
  s->c = *c;
  offsets(c, s->off);
  const int P = s->off[12];
  s->params = (float*)calloc(P, 4); s->grads = (float*)calloc(P, 4); s->m = (float*)calloc(P, 4); s->v = (float*)calloc(P, 4);
  for (int a = 0; a < 12; ++a) { s->betap[2 * a] = 0.9; s->betap[2 * a + 1] = 0.999; }
  s->cap = 2 * c->min_replay_size;                                       /* a2c.jl:46 */
  s->rb_state = (double*)calloc((size_t)s->cap * c->obs_dim, 8); s->rb_action = (int32_t*)calloc(s->cap, 4);
  s->rb_reward = (double*)calloc(s->cap, 8); s->rb_terminal = (uint8_t*)calloc(s->cap, 1);
  s->ptr = 0; s->size = 0;
  env_reset(c, s->env, 0, 2); s->env_t = 0;                              /* a2c.jl:52 reset!(env) */
  s->global_step = 0;
  return s;
}
void a2c_destroy(a2c_state* s) {
  if (!s) return;
  free(s->params); free(s->grads); free(s->m); free(s->v);
  free(s->rb_state); free(s->rb_action); free(s->rb_reward); free(s->rb_terminal); free(s);
}
int32_t a2c_param_count(const a2c_state* s) { return s->off[12]; }
void a2c_set_params(a2c_state* s, const float* p) { memcpy(s->params, p, sizeof(float) * s->off[12]); }
void a2c_get_params(const a2c_state* s, float* p) { memcpy(p, s->params, sizeof(float) * s->off[12]); }
void a2c_get_env(const a2c_state* s, double* state4, int64_t* global_step, int32_t* rb_size) {
  memcpy(state4, s->env, sizeof(double) * 4);
  *global_step = s->global_step; *rb_size = s->size;
}
void a2c_get_buffer(const a2c_state* s, double* state, int32_t* action, double* reward, uint8_t* terminal) {
  memcpy(state, s->rb_state, sizeof(double) * (size_t)s->size * s->c.obs_dim);
  memcpy(action, s->rb_action, sizeof(int32_t) * s->size);
  memcpy(reward, s->rb_reward, sizeof(double) * s->size);
  memcpy(terminal, s->rb_terminal, s->size);
}

int64_t a2c_run_until_update(a2c_state* s, int64_t max_env_steps, a2c_train_stats* ts, a2c_episode* eps, int32_t max_eps,
                             int32_t* n_eps) {
  const a2c_config* c = &s->c;
  const int d = c->obs_dim, A = c->n_act;
  int64_t taken = 0;
  *n_eps = 0;
  ts->trained = 0; ts->n = 0; ts->actor_loss = 0.0; ts->critic_loss = 0.0;
  while (taken < max_env_steps && s->global_step < c->total_timesteps) {
    s->global_step += 1;                                                  /* a2c.jl:53 for global_step in 1:total */
    taken += 1;
    const uint64_t gstep = (uint64_t)s->global_step;
    double obs[64], z[16], p[16];
    memcpy(obs, s->env, sizeof(double) * d);                              /* a2c.jl:55 deepcopy(state(env)) */
    a2c_forward(c, s->params, 0, obs, z);
    softmax64(z, A, p);                                                   /* a2c.jl:56 */
    const int action = sample_categorical(p, A, orc_u53(c->seed, 0, gstep, 0));   /* a2c.jl:57-58 */
    int32_t done;
    a2c_cartpole_step(s->env, &s->env_t, action, c->max_steps, &done);   /* a2c.jl:60 */
    const double rew = done ? 0.0 : 1.0;                                  /* reward(env), Q12 */
    memcpy(s->rb_state + (size_t)d * s->ptr, obs, sizeof(double) * d);   /* a2c.jl:62-68 Buffer.add! */
    s->rb_action[s->ptr] = action; s->rb_reward[s->ptr] = rew; s->rb_terminal[s->ptr] = (uint8_t)done;
    s->ptr = s->ptr + 1 >= s->cap ? 0 : s->ptr + 1;
    s->size = s->size + 1 > s->cap ? s->cap : s->size + 1;
    s->episode_return += rew; s->episode_length += 1;                     /* a2c.jl:71-72 */
    if (done) {                                                           /* a2c.jl:74 */
      int trained = 0;
      if (s->size > c->min_replay_size) {                                 /* a2c.jl:75 */
        const int n = s->size;
        double fv;
        a2c_forward(c, s->params, 1, s->env, &fv);                        /* a2c.jl:78 critic(state(env))[1] */
        double* G = (double*)malloc(sizeof(double) * n);
        double* adv = (double*)malloc(sizeof(double) * n);
        a2c_discounted_future_rewards(s->rb_reward, s->rb_terminal, n, fv, c->gamma, G);   /* a2c.jl:79 */
        a2c_loss_grads(c, s->params, s->rb_state, s->rb_action, G, n, s->grads, &ts->critic_loss, &ts->actor_loss, adv);
        clipnorm_adam_range(s->off, 6, 12, s->params, s->grads, s->m, s->v, s->betap, c->lr);   /* a2c.jl:88 */
        clipnorm_adam_range(s->off, 0, 6, s->params, s->grads, s->m, s->v, s->betap, c->lr);    /* a2c.jl:98 */
        free(G); free(adv);
        ts->n = n; ts->trained = 1; trained = 1;
        s->size = 0; s->ptr = 0;                                          /* a2c.jl:102 Buffer.clear! */
      }
      if (*n_eps < max_eps) {                                             /* a2c.jl:105-106 */
        eps[*n_eps].episode_return = s->episode_return; eps[*n_eps].episode_length = s->episode_length;
        eps[*n_eps].global_step = s->global_step;
        *n_eps += 1;
      }
      s->episode_length = 0; s->episode_return = 0.0;                     /* a2c.jl:108 */
      env_reset(c, s->env, gstep, 1); s->env_t = 0;                       /* a2c.jl:109 */
      if (trained) break;
    }
  }
  return taken;
}
