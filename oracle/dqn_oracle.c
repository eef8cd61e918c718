/*
 * dqn_oracle.c — CPU restatement of src/algorithms/dqn.jl (TEST INFRASTRUCTURE ONLY). See dqn_oracle.h.
 */
#include "dqn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "a2c_oracle.h" /* a2c_cartpole_step: CartPoleEnv{Float64} */
#include "ppo_oracle.h" /* orc_philox, orc_u53 */

enum { oW1 = 0, ob1 = DQN_H1 * DQN_D, oW2 = ob1 + DQN_H1, ob2 = oW2 + DQN_H2 * DQN_H1, oW3 = ob2 + DQN_H2, ob3 = oW3 + DQN_A * DQN_H2 };
static const int OFF[7] = {oW1, ob1, oW2, ob2, oW3, ob3, DQN_P};

double dqn_linear_schedule(double start_e, double end_e, double duration, double t) {
  const double slope = (end_e - start_e) / duration;
  const double v = slope * t + start_e;
  return v > end_e ? v : end_e;
}

/* Chain(Dense(4,120,relu), Dense(120,84,relu), Dense(84,2)) on a Float64 vector; Float64 accumulation in k order */
static void forward_h(const float* p, const double* x, double* h1, double* h2, double* q) {
  for (int i = 0; i < DQN_H1; ++i) {
    double acc = 0.0;
    for (int k = 0; k < DQN_D; ++k) acc += (double)p[oW1 + i + DQN_H1 * k] * x[k];
    acc += (double)p[ob1 + i];
    h1[i] = acc > 0.0 ? acc : 0.0;
  }
  for (int j = 0; j < DQN_H2; ++j) {
    double acc = 0.0;
    for (int k = 0; k < DQN_H1; ++k) acc += (double)p[oW2 + j + DQN_H2 * k] * h1[k];
    acc += (double)p[ob2 + j];
    h2[j] = acc > 0.0 ? acc : 0.0;
  }
  for (int a = 0; a < DQN_A; ++a) {
    double acc = 0.0;
    for (int k = 0; k < DQN_H2; ++k) acc += (double)p[oW3 + a + DQN_A * k] * h2[k];
    q[a] = acc + (double)p[ob3 + a];
  }
}
void dqn_forward(const float* params, const double* x, double* q) {
  double h1[DQN_H1], h2[DQN_H2];
  forward_h(params, x, h1, h2, q);
}

double dqn_loss_grads(const float* qp, const float* tp, const double* state, const double* next_state, const int32_t* action,
                      const double* reward, const uint8_t* terminal, int32_t n, double gamma, float* grads) {
  double* g = (double*)calloc(DQN_P, sizeof(double));
  double* h1 = (double*)malloc(sizeof(double) * DQN_H1 * (size_t)n);
  double* h2 = (double*)malloc(sizeof(double) * DQN_H2 * (size_t)n);
  double* dz = (double*)malloc(sizeof(double) * (size_t)n);
  double* d2 = (double*)malloc(sizeof(double) * DQN_H2 * (size_t)n);
  double* d1 = (double*)malloc(sizeof(double) * DQN_H1 * (size_t)n);
  double loss = 0.0;
  for (int b = 0; b < n; ++b) {
    double t1[DQN_H1], t2[DQN_H2], tq[DQN_A], q[DQN_A];
    forward_h(tp, next_state + (size_t)DQN_D * b, t1, t2, tq);
    const double next_q = tq[1] > tq[0] ? tq[1] : tq[0];                    /* eachcol .|> maximum (dqn.jl:99) */
    const double td = reward[b] + gamma * next_q * (1.0 - (double)terminal[b]);   /* dqn.jl:100 */
    forward_h(qp, state + (size_t)DQN_D * b, h1 + (size_t)DQN_H1 * b, h2 + (size_t)DQN_H2 * b, q);
    const double diff = td - q[action[b]];                                  /* Flux.mse(td_target, q) (dqn.jl:107) */
    loss += diff * diff;
    dz[b] = -2.0 * diff / (double)n;
  }
  loss = loss / (double)n;
  /* pullbacks; every sum over samples runs in sample order */
  for (int b = 0; b < n; ++b) {
    const int a = action[b];
    for (int j = 0; j < DQN_H2; ++j) {
      const double s = (double)qp[oW3 + a + DQN_A * j] * dz[b];
      d2[(size_t)DQN_H2 * b + j] = h2[(size_t)DQN_H2 * b + j] > 0.0 ? s : 0.0;
    }
    for (int k = 0; k < DQN_H1; ++k) {
      double s = 0.0;
      for (int j = 0; j < DQN_H2; ++j) s += (double)qp[oW2 + j + DQN_H2 * k] * d2[(size_t)DQN_H2 * b + j];
      d1[(size_t)DQN_H1 * b + k] = h1[(size_t)DQN_H1 * b + k] > 0.0 ? s : 0.0;
    }
  }
  for (int b = 0; b < n; ++b) {
    const int a = action[b];
    g[ob3 + a] += dz[b];
    for (int j = 0; j < DQN_H2; ++j) g[oW3 + a + DQN_A * j] += dz[b] * h2[(size_t)DQN_H2 * b + j];
    for (int j = 0; j < DQN_H2; ++j) {
      const double dj = d2[(size_t)DQN_H2 * b + j];
      g[ob2 + j] += dj;
      for (int k = 0; k < DQN_H1; ++k) g[oW2 + j + DQN_H2 * k] += dj * h1[(size_t)DQN_H1 * b + k];
    }
    for (int i = 0; i < DQN_H1; ++i) {
      const double di = d1[(size_t)DQN_H1 * b + i];
      g[ob1 + i] += di;
      for (int k = 0; k < DQN_D; ++k) g[oW1 + i + DQN_H1 * k] += di * state[(size_t)DQN_D * b + k];
    }
  }
  for (int i = 0; i < DQN_P; ++i) grads[i] = (float)g[i];
  free(g); free(h1); free(h2); free(dz); free(d2); free(d1);
  return loss;
}

/* k distinct indices of [0, n): candidates c = 0, 1, 2, … are floor(u32 * n / 2^32) from Philox(ctr = (c, gstep), stream 0xD9);
 * a candidate already chosen is skipped. Uniform over k-permutations. */
void dqn_sample_indices(uint64_t seed, uint64_t gstep, int32_t n, int32_t k, int32_t* out) {
  uint8_t* used = (uint8_t*)calloc((size_t)n, 1);
  int got = 0;
  for (uint32_t c = 0; got < k; ++c) {
    uint32_t o[4];
    orc_philox(c, (uint32_t)gstep, (uint32_t)(gstep >> 32), 0xD9u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const uint32_t j = (uint32_t)(((uint64_t)o[0] * (uint64_t)n) >> 32);
    if (!used[j]) { used[j] = 1; out[got++] = (int32_t)j; }
  }
  free(used);
}

/* Flux 0.13.4 Adam(η) without ClipNorm (dqn.jl:41,109), per array β powers */
static void adam(float* params, const float* grads, float* m, float* v, double* betap, double eta) {
  const double b1 = 0.9, b2 = 0.999, epsn = 1e-8;
  for (int a = 0; a < 6; ++a) {
    double* bp = betap + 2 * a;
    for (int i = OFF[a]; i < OFF[a + 1]; ++i) {
      const double gg = grads[i];
      m[i] = (float)(b1 * (double)m[i] + (1 - b1) * gg);
      v[i] = (float)(b2 * (double)v[i] + (1 - b2) * gg * gg);
      const double delta = (double)m[i] / (1 - bp[0]) / (sqrt((double)v[i] / (1 - bp[1])) + epsn) * eta;
      params[i] = params[i] - (float)delta;
    }
    bp[0] *= b1; bp[1] *= b2;
  }
}

struct dqn_state {
  dqn_config c;
  float q[DQN_P], t[DQN_P], grads[DQN_P], m[DQN_P], v[DQN_P];
  double betap[12];
  double env[4]; int32_t env_t;
  int64_t cap, ptr, size;
  double *rb_state, *rb_next; int32_t* rb_action; double* rb_reward; uint8_t* rb_terminal;
  double episode_return; int64_t episode_length, global_step, n_updates;
  double last_loss;
};

static void env_reset(const dqn_config* c, double* s, uint64_t gstep, uint32_t stream) {
  for (int i = 0; i < 4; ++i) s[i] = 0.1 * orc_u53(c->seed, (uint32_t)i, gstep, stream) - 0.05;
}

dqn_state* dqn_create(const dqn_config* c) {
  dqn_state* s = (dqn_state*)calloc(1, sizeof(dqn_state));
  s->c = *c;
  for (int a = 0; a < 6; ++a) { s->betap[2 * a] = 0.9; s->betap[2 * a + 1] = 0.999; }
  s->cap = c->buffer_size;
  s->rb_state = (double*)calloc((size_t)s->cap * DQN_D, 8); s->rb_next = (double*)calloc((size_t)s->cap * DQN_D, 8);
  s->rb_action = (int32_t*)calloc((size_t)s->cap, 4); s->rb_reward = (double*)calloc((size_t)s->cap, 8);
  s->rb_terminal = (uint8_t*)calloc((size_t)s->cap, 1);
  env_reset(c, s->env, 0, 2);                                            /* dqn.jl:56 reset!(env) */
  return s;
}
void dqn_destroy(dqn_state* s) {
  if (!s) return;
  free(s->rb_state); free(s->rb_next); free(s->rb_action); free(s->rb_reward); free(s->rb_terminal); free(s);
}
void dqn_set_params(dqn_state* s, const float* p) { memcpy(s->q, p, sizeof(s->q)); memcpy(s->t, p, sizeof(s->t)); }
void dqn_get_params(const dqn_state* s, float* q, float* t) { memcpy(q, s->q, sizeof(s->q)); if (t) memcpy(t, s->t, sizeof(s->t)); }
void dqn_get_env(const dqn_state* s, double* state4, int64_t* global_step, int64_t* rb_size, double* last_loss, int64_t* n_updates) {
  memcpy(state4, s->env, 32);
  *global_step = s->global_step; *rb_size = s->size; *last_loss = s->last_loss; *n_updates = s->n_updates;
}

int64_t dqn_run(dqn_state* s, int64_t max_env_steps, dqn_episode* eps, int32_t max_eps, int32_t* n_eps,
                dqn_loss_record* losses, int32_t max_losses, int32_t* n_losses) {
  const dqn_config* c = &s->c;
  int64_t taken = 0;
  *n_eps = 0; *n_losses = 0;
  const int k = (int)c->batch_size;
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
  double* bs = (double*)malloc(sizeof(double) * DQN_D * (size_t)k);
  double* bn = (double*)malloc(sizeof(double) * DQN_D * (size_t)k);
  int32_t* ba = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
  double* br = (double*)malloc(sizeof(double) * (size_t)k);
  uint8_t* bt = (uint8_t*)malloc((size_t)k);
  while (taken < max_env_steps && s->global_step < c->total_timesteps) {
    s->global_step += 1; taken += 1;                                      /* dqn.jl:57 */
    const uint64_t gstep = (uint64_t)s->global_step;
    double obs[4];
    memcpy(obs, s->env, 32);                                              /* dqn.jl:58 */
    const double eps_t = dqn_linear_schedule(c->epsilon_start, c->epsilon_end, c->epsilon_duration, (double)s->global_step);
    int action;
    if (orc_u53(c->seed, 0, gstep, 0) < eps_t) {                          /* dqn.jl:61-62 rand() < ϵ → rand(action_space) */
      uint32_t o[4];
      orc_philox(0, (uint32_t)gstep, (uint32_t)(gstep >> 32), 4u, (uint32_t)c->seed, (uint32_t)(c->seed >> 32), o);
      action = (int)(o[0] >> 31);
    } else {
      double q[DQN_A];
      dqn_forward(s->q, obs, q);                                          /* dqn.jl:64-65 argmax(qs): first maximum */
      action = q[1] > q[0] ? 1 : 0;
    }
    int32_t done;
    a2c_cartpole_step(s->env, &s->env_t, action, c->max_steps, &done);   /* dqn.jl:68 */
    const double rew = done ? 0.0 : 1.0;
    memcpy(s->rb_state + DQN_D * s->ptr, obs, 32);                        /* dqn.jl:71-78 Buffer.add! */
    memcpy(s->rb_next + DQN_D * s->ptr, s->env, 32);
    s->rb_action[s->ptr] = action; s->rb_reward[s->ptr] = rew; s->rb_terminal[s->ptr] = (uint8_t)done;
    s->ptr = s->ptr + 1 >= s->cap ? 0 : s->ptr + 1;
    s->size = s->size + 1 > s->cap ? s->cap : s->size + 1;
    s->episode_return += rew; s->episode_length += 1;                     /* dqn.jl:81-82 */
    if (done) {                                                           /* dqn.jl:83-90 */
      if (*n_eps < max_eps) {
        eps[*n_eps].episode_return = s->episode_return; eps[*n_eps].episode_length = s->episode_length;
        eps[*n_eps].global_step = s->global_step; eps[*n_eps].epsilon = eps_t;
        *n_eps += 1;
      }
      s->episode_length = 0; s->episode_return = 0.0;
      env_reset(c, s->env, gstep, 1); s->env_t = 0;
    }
    if (s->global_step > c->min_buff_size && s->global_step % c->train_freq == 0) {   /* dqn.jl:93 */
      dqn_sample_indices(c->seed, gstep, (int32_t)s->size, k, idx);       /* dqn.jl:94 Buffer.sample(rb, batch_size) */
      for (int b = 0; b < k; ++b) {
        memcpy(bs + DQN_D * b, s->rb_state + DQN_D * (size_t)idx[b], 32);
        memcpy(bn + DQN_D * b, s->rb_next + DQN_D * (size_t)idx[b], 32);
        ba[b] = s->rb_action[idx[b]]; br[b] = s->rb_reward[idx[b]]; bt[b] = s->rb_terminal[idx[b]];
      }
      s->last_loss = dqn_loss_grads(s->q, s->t, bs, bn, ba, br, bt, k, c->gamma, s->grads);   /* dqn.jl:96-108 */
      adam(s->q, s->grads, s->m, s->v, s->betap, c->lr);                  /* dqn.jl:109 */
      s->n_updates += 1;
      if (s->global_step % c->target_net_freq == 0) memcpy(s->t, s->q, sizeof(s->q));   /* dqn.jl:111-113 */
      if (s->global_step % c->log_frequency == 0 && *n_losses < max_losses) {           /* dqn.jl:115-117 */
        losses[*n_losses].global_step = s->global_step; losses[*n_losses].loss = s->last_loss;
        *n_losses += 1;
      }
    }
  }
  free(idx); free(bs); free(bn); free(ba); free(br); free(bt);
  return taken;
}
