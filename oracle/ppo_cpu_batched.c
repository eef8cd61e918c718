/*
 * ppo_cpu_batched.c — a THROUGHPUT-oriented CPU PPO iteration (TEST / MEASUREMENT INFRASTRUCTURE ONLY, like ppo_oracle.c).
 *
 * Why it exists: bench.py's `cpu_baseline.value` times ppo_oracle.c, a scalar restatement that keeps the reference's operation
 * order (per-sample strided GEMV, -O2 -ffp-contract=off) — right for parity, one to two orders of magnitude slower than what
 * the reference itself (Julia / Flux on BLAS, ppo.jl:127-128,202-250) would do on the same cores. This file is the same
 * algorithm — same loop structure as orc_iterate (ppo.jl:117-253), same env, sampler, GAE, loss closure, per-array
 * ClipNorm + Adam — with the network passes BATCHED: samples are processed in blocks of 64 in a [feature][sample] layout so the
 * compiler vectorises over samples (AVX-512 / AVX2 FMAs with a broadcast weight), gradients accumulate per thread and are
 * folded once per minibatch. Built with -O3 -march=native on the box that runs it (oracle/Makefile target `batched`).
 * It is NOT the parity oracle: operation order differs (float32 block sums), so results agree with ppo_oracle.c to ~1e-5,
 * not bit for bit (tests/test_oracle.py::test_batched_cpu_iteration_tracks_the_oracle). `cpu_baseline.batched` reports it.
 *
 * Reuses from ppo_oracle.c (linked into the same shared object): orc_state, orc_cartpole_step, orc_env_reset, orc_u53,
 * orc_gae_batch, orc_clipnorm_adam, orc_param_offsets.
 */
#include "ppo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BS 64          /* samples per block */
#define HMAX 64        /* this fast path covers the reference's own shape: hidden 64 (networks.jl:36), obs <= 8, act <= 4 */

typedef struct { float v[BS]; } row_t;   /* one feature of a block of samples */

static inline void tanh_rows(row_t* x, int n) {   /* NNlib tanh_fast (networks.jl:6), vectorised over the block */
  for (int r = 0; r < n; ++r)
    for (int s = 0; s < BS; ++s) {
      float t = x[r].v[s], x2 = t * t;
      float nn = x2 * (x2 * (x2 * (x2 * 1.587199e-8f + 2.2332108e-5f) + 0.0035974074f) + 0.1346604f) + 1.0f;
      float dd = x2 * (x2 * (x2 * (x2 * 8.7767893e-7f + 0.0003453992f) + 0.026262015f) + 0.4679937f) + 1.0f;
      float y = t * (nn / dd);
      x[r].v[s] = x2 < 66.0f ? y : (t > 0.0f ? 1.0f : -1.0f);
    }
}

/* out[o][s] = b[o] + Σ_i W[o + no·i]·in[i][s]   (W is (out, in) column-major, like Flux). Four outputs per pass: a row of `in`
 * is loaded once for four FMAs (the plain one-output loop is bound by those loads). */
static void dense_fwd(const float* W, const float* b, int no, int ni, const row_t* in, row_t* out) {
  int o = 0;
  for (; o + 4 <= no; o += 4) {
    float a0[BS], a1[BS], a2[BS], a3[BS];
    for (int s = 0; s < BS; ++s) { a0[s] = b[o]; a1[s] = b[o + 1]; a2[s] = b[o + 2]; a3[s] = b[o + 3]; }
    for (int i = 0; i < ni; ++i) {
      const float w0 = W[o + no * i], w1 = W[o + 1 + no * i], w2 = W[o + 2 + no * i], w3 = W[o + 3 + no * i];
      for (int s = 0; s < BS; ++s) {
        const float x = in[i].v[s];
        a0[s] += w0 * x; a1[s] += w1 * x; a2[s] += w2 * x; a3[s] += w3 * x;
      }
    }
    memcpy(out[o].v, a0, sizeof(a0)); memcpy(out[o + 1].v, a1, sizeof(a1)); memcpy(out[o + 2].v, a2, sizeof(a2)); memcpy(out[o + 3].v, a3, sizeof(a3));
  }
  for (; o < no; ++o) {
    float acc[BS];
    for (int s = 0; s < BS; ++s) acc[s] = b[o];
    for (int i = 0; i < ni; ++i) {
      const float w = W[o + no * i];
      for (int s = 0; s < BS; ++s) acc[s] += w * in[i].v[s];
    }
    memcpy(out[o].v, acc, sizeof(acc));
  }
}
/* din[i][s] = Σ_o W[o + no·i]·dout[o][s] */
static void dense_bwd_data(const float* W, int no, int ni, const row_t* dout, row_t* din) {
  int i = 0;
  for (; i + 4 <= ni; i += 4) {
    float a0[BS], a1[BS], a2[BS], a3[BS];
    for (int s = 0; s < BS; ++s) { a0[s] = 0.0f; a1[s] = 0.0f; a2[s] = 0.0f; a3[s] = 0.0f; }
    for (int o = 0; o < no; ++o) {
      const float w0 = W[o + no * i], w1 = W[o + no * (i + 1)], w2 = W[o + no * (i + 2)], w3 = W[o + no * (i + 3)];
      for (int s = 0; s < BS; ++s) {
        const float x = dout[o].v[s];
        a0[s] += w0 * x; a1[s] += w1 * x; a2[s] += w2 * x; a3[s] += w3 * x;
      }
    }
    memcpy(din[i].v, a0, sizeof(a0)); memcpy(din[i + 1].v, a1, sizeof(a1)); memcpy(din[i + 2].v, a2, sizeof(a2)); memcpy(din[i + 3].v, a3, sizeof(a3));
  }
  for (; i < ni; ++i) {
    float acc[BS];
    for (int s = 0; s < BS; ++s) acc[s] = 0.0f;
    for (int o = 0; o < no; ++o) {
      const float w = W[o + no * i];
      for (int s = 0; s < BS; ++s) acc[s] += w * dout[o].v[s];
    }
    memcpy(din[i].v, acc, sizeof(acc));
  }
}
/* gW[o + no·i] += Σ_s dout[o][s]·in[i][s];  gb[o] += Σ_s dout[o][s].  The reduction runs over the sample axis: the block of `in` is
 * transposed once so that the vector axis is the OUTPUT column (no horizontal sums in the inner loop). */
static void dense_bwd_weight(int no, int ni, const row_t* dout, const row_t* in, float* gW, float* gb) {
  if (no == HMAX && ni == HMAX) {
    static _Thread_local float inT[BS][HMAX];
    static _Thread_local float acc[HMAX][HMAX];   /* acc[i][o] */
    for (int i = 0; i < ni; ++i)
      for (int s = 0; s < BS; ++s) inT[s][i] = in[i].v[s];
    memset(acc, 0, sizeof(acc));
    for (int o = 0; o < HMAX; o += 4)
      for (int s = 0; s < BS; ++s) {
        const float d0 = dout[o].v[s], d1 = dout[o + 1].v[s], d2 = dout[o + 2].v[s], d3 = dout[o + 3].v[s];
        for (int i = 0; i < HMAX; ++i) {
          const float x = inT[s][i];
          acc[o][i] += d0 * x; acc[o + 1][i] += d1 * x; acc[o + 2][i] += d2 * x; acc[o + 3][i] += d3 * x;
        }
      }
    for (int o = 0; o < HMAX; ++o)
      for (int i = 0; i < HMAX; ++i) gW[o + no * i] += acc[o][i];
  } else {
    for (int o = 0; o < no; ++o)
      for (int i = 0; i < ni; ++i) {
        float t = 0.0f;
        for (int s = 0; s < BS; ++s) t += dout[o].v[s] * in[i].v[s];
        gW[o + no * i] += t;
      }
  }
  for (int o = 0; o < no; ++o) {
    float t = 0.0f;
    for (int s = 0; s < BS; ++s) t += dout[o].v[s];
    gb[o] += t;
  }
}

typedef struct { row_t x[8], h1[HMAX], h2[HMAX], out[4]; } net_acts;

static void net_forward(const orc_config* c, const float* p, const int32_t* o, int net, net_acts* a) {
  const int h = c->hidden, base = net ? 6 : 0, no = net ? 1 : c->n_act;
  dense_fwd(p + o[base], p + o[base + 1], h, c->obs_dim, a->x, a->h1); tanh_rows(a->h1, h);
  dense_fwd(p + o[base + 2], p + o[base + 3], h, h, a->h1, a->h2); tanh_rows(a->h2, h);
  dense_fwd(p + o[base + 4], p + o[base + 5], no, h, a->h2, a->out);
}
static void net_backward(const orc_config* c, const float* p, const int32_t* o, int net, net_acts* a, const row_t* dout, float* g) {
  const int h = c->hidden, base = net ? 6 : 0, no = net ? 1 : c->n_act;
  static _Thread_local row_t d2[HMAX], d1[HMAX];
  dense_bwd_weight(no, h, dout, a->h2, g + o[base + 4], g + o[base + 5]);
  dense_bwd_data(p + o[base + 4], no, h, dout, d2);
  for (int i = 0; i < h; ++i) for (int s = 0; s < BS; ++s) d2[i].v[s] *= 1.0f - a->h2[i].v[s] * a->h2[i].v[s];
  dense_bwd_weight(h, h, d2, a->h1, g + o[base + 2], g + o[base + 3]);
  dense_bwd_data(p + o[base + 2], h, h, d2, d1);
  for (int i = 0; i < h; ++i) for (int s = 0; s < BS; ++s) d1[i].v[s] *= 1.0f - a->h1[i].v[s] * a->h1[i].v[s];
  dense_bwd_weight(h, c->obs_dim, d1, a->x, g + o[base], g + o[base + 1]);
}

static void softmax2(const float* z, int A, float* p, float* lp) {   /* NNlib softmax / logsoftmax over one column */
  float m = z[0];
  for (int a = 1; a < A; ++a) m = z[a] > m ? z[a] : m;
  float s = 0.0f;
  for (int a = 0; a < A; ++a) { p[a] = expf(z[a] - m); s += p[a]; }
  const float l = logf(s);
  for (int a = 0; a < A; ++a) { lp[a] = z[a] - m - l; p[a] = p[a] / s; }
}

/* ppo.jl:123-166, one thread per block of 64 envs for all num_steps steps (envs are independent) */
static void batched_rollout(const orc_config* c, orc_state* s) {
  const int nt = c->num_envs, k = c->num_steps, d = c->obs_dim, A = c->n_act;
  int32_t o[13];
  orc_param_offsets(c, o);
  double epc = 0, eprs = 0, epls = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : epc, eprs, epls)
  for (int e0 = 0; e0 < nt; e0 += BS) {
    static _Thread_local net_acts act, cri;
    const int ne = nt - e0 < BS ? nt - e0 : BS;
    for (int t = 0; t < k; ++t) {
      const uint64_t gstep = s->iteration * (uint64_t)k + (uint64_t)t;
      for (int i = 0; i < d; ++i)
        for (int q = 0; q < BS; ++q) act.x[i].v[q] = cri.x[i].v[q] = q < ne ? s->cur_obs[(size_t)d * (e0 + q) + i] : 0.0f;
      net_forward(c, s->params, o, 0, &act);
      net_forward(c, s->params, o, 1, &cri);
      for (int q = 0; q < ne; ++q) {
        const int e = e0 + q;
        const uint32_t gid = (uint32_t)(c->env_id_offset + e);
        const size_t b = (size_t)e + (size_t)nt * t;
        float z[4], p[4], lp[4];
        for (int a = 0; a < A; ++a) z[a] = act.out[a].v[q];
        softmax2(z, A, p, lp);
        float sw = 0.0f;
        for (int a = 0; a < A; ++a) sw += p[a];
        const double tt = orc_u53(c->seed, gid, gstep, 0) * (double)sw;   /* StatsBase.sample (ppo.jl:26) */
        int a = 0; float cw = p[0];
        while ((double)cw < tt && a < A - 1) { a += 1; cw += p[a]; }
        float* es = s->env_state + (size_t)d * e;
        float* co = s->cur_obs + (size_t)d * e;
        int32_t done;
        s->ep_length[e] += 1;
        orc_cartpole_step(es, &s->env_t[e], a, 500, &done);
        const float rew = done ? 0.0f : 1.0f;
        memcpy(s->obs + b * d, co, sizeof(float) * d);
        s->action[b] = a; s->logprob[b] = lp[a]; s->reward[b] = rew; s->terminal[b] = s->next_done[e]; s->value[b] = cri.out[0].v[q];
        memcpy(co, es, sizeof(float) * d);
        s->next_done[e] = (uint8_t)done;
        s->ep_return[e] += rew;
        if (done) {
          epc += 1; eprs += s->ep_return[e]; epls += s->ep_length[e];
          s->ep_return[e] = 0.0f; s->ep_length[e] = 0;
          orc_env_reset(c, es, gid, gstep, 1); s->env_t[e] = 0;
          if (!c->stale_obs) memcpy(co, es, sizeof(float) * d);
        }
      }
    }
  }
  s->ep_count = epc; s->ep_return_sum = eprs; s->ep_length_sum = epls;
}

/* ppo.jl:202-250 for one minibatch: loss closure + gradient, batched; then per-array ClipNorm + Adam */
static void batched_update(const orc_config* c, orc_state* s, const int32_t* mb_inds, int M, double eta, orc_stats* st) {
  int32_t o[13];
  orc_param_offsets(c, o);
  const int P = o[12], A = c->n_act, d = c->obs_dim;
  const float eps = c->clip_coef, lo = 1 - eps, hi = 1 + eps;
  double sum = 0.0, sq = 0.0;
#pragma omp parallel for reduction(+ : sum, sq)
  for (int j = 0; j < M; ++j) { const double a = s->adv[mb_inds[j]]; sum += a; sq += a * a; }
  const double mean_d = sum / M;
  double var_d = (sq - M * mean_d * mean_d) / (M - 1);
  if (var_d < 0) var_d = 0;
  const float mean_f = (float)mean_d, std_f = (float)sqrt(var_d);
  const double denom = (double)std_f + 1e-8;
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
#endif
  const int nblk = (M + BS - 1) / BS;
  if (nth > nblk) nth = nblk;
  float* gacc = (float*)calloc((size_t)P * nth, sizeof(float));
  float* newv = (float*)malloc(sizeof(float) * (size_t)M);
  /* pass 1 (critic only): u = mean(newvalue .- returns.^2), ppo.jl:232 (Q4) */
  double usum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : usum) num_threads(nth)
  for (int blk = 0; blk < nblk; ++blk) {
    static _Thread_local net_acts cri;
    const int j0 = blk * BS, nj = M - j0 < BS ? M - j0 : BS;
    for (int i = 0; i < d; ++i) for (int q = 0; q < BS; ++q) cri.x[i].v[q] = q < nj ? s->obs[(size_t)d * mb_inds[j0 + q] + i] : 0.0f;
    net_forward(c, s->params, o, 1, &cri);
    for (int q = 0; q < nj; ++q) { const float v = cri.out[0].v[q], R = s->ret[mb_inds[j0 + q]]; newv[j0 + q] = v; usum += (double)(v - R * R); }
  }
  const float u = (float)(usum / M);
  double nwin = 0.0;
  if (c->clip_value_loss && u > 0.0f) {
    for (int j = 0; j < M; ++j) {
      const int sidx = mb_inds[j];
      float dv = newv[j] - s->value[sidx];
      float cl = dv < -eps ? -eps : (dv > eps ? eps : dv);
      float vc = s->value[sidx] + cl, q = (vc - s->ret[sidx]) * (vc - s->ret[sidx]);
      if (u > q) nwin += 1.0;
    }
  }
  double pg_sum = 0.0, vmax_sum = 0.0, ent_sum = 0.0;
#pragma omp parallel reduction(+ : pg_sum, vmax_sum, ent_sum) num_threads(nth)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    float* g = gacc + (size_t)P * tid;
    static _Thread_local net_acts act, cri;
    static _Thread_local row_t dz[4], dv;
#pragma omp for schedule(static)
    for (int blk = 0; blk < nblk; ++blk) {
      const int j0 = blk * BS, nj = M - j0 < BS ? M - j0 : BS;
      for (int i = 0; i < d; ++i)
        for (int q = 0; q < BS; ++q) act.x[i].v[q] = cri.x[i].v[q] = q < nj ? s->obs[(size_t)d * mb_inds[j0 + q] + i] : 0.0f;
      net_forward(c, s->params, o, 0, &act);
      net_forward(c, s->params, o, 1, &cri);
      for (int q = 0; q < BS; ++q) {
        for (int a = 0; a < A; ++a) dz[a].v[q] = 0.0f;
        dv.v[q] = 0.0f;
        if (q >= nj) continue;
        const int sidx = mb_inds[j0 + q];
        float z[4], p[4], lp[4];
        for (int a = 0; a < A; ++a) z[a] = act.out[a].v[q];
        softmax2(z, A, p, lp);
        const int a_s = s->action[sidx];
        double H = 0.0;
        for (int a = 0; a < A; ++a) { const float e = -(p[a] * lp[a]); ent_sum += e; H += e; }
        const double Ahat = (double)(s->adv[sidx] - mean_f) / denom;
        const float ratio = expf(lp[a_s] - s->logprob[sidx]);
        const float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
        const double pg1 = -Ahat * (double)ratio, pg2 = -Ahat * (double)rc;
        double dnlp;
        if (pg1 > pg2) { pg_sum += pg1; dnlp = pg1; } else { pg_sum += pg2; dnlp = (ratio >= lo && ratio <= hi) ? pg1 : 0.0; }
        dnlp /= M;
        for (int a = 0; a < A; ++a)
          dz[a].v[q] = (float)(dnlp * ((a == a_s ? 1.0 : 0.0) - (double)p[a]) + (double)c->ent_coeff / ((double)A * M) * (double)p[a] * ((double)lp[a] + H));
        const float v = cri.out[0].v[q], R = s->ret[sidx];
        double dvv_;
        if (c->clip_value_loss) {
          const float dvv = v - s->value[sidx];
          const float cl = dvv < -eps ? -eps : (dvv > eps ? eps : dvv);
          const float vc = s->value[sidx] + cl, qq = (vc - R) * (vc - R);
          const int q_wins = !(u > qq);
          vmax_sum += q_wins ? (double)qq : (double)u;
          const double inner = (q_wins && dvv >= -eps && dvv <= eps) ? 2.0 * (double)(vc - R) : 0.0;
          dvv_ = (double)c->v_coef * 0.5 / M * (nwin / M + inner);
        } else {
          const float e = v - R;
          vmax_sum += (double)(e * e);
          dvv_ = (double)c->v_coef * 0.5 / M * 2.0 * (double)e;
        }
        dv.v[q] = (float)dvv_;
      }
      net_backward(c, s->params, o, 0, &act, dz, g);
      net_backward(c, s->params, o, 1, &cri, &dv, g);
    }
  }
#pragma omp parallel for
  for (int i = 0; i < P; ++i) {
    double t = 0.0;
    for (int th = 0; th < nth; ++th) t += gacc[(size_t)P * th + i];
    s->grads[i] = (float)t;
  }
  free(gacc); free(newv);
  if (st) {
    st->pg_loss = pg_sum / M;
    st->v_loss = 0.5 * (double)(float)(vmax_sum / M);
    st->entropy_loss = (double)(float)(ent_sum / ((double)A * M));
    st->loss = st->pg_loss - (double)(c->ent_coeff * (float)st->entropy_loss) + (double)c->v_coef * st->v_loss;
    st->adv_mean = mean_f; st->adv_std = std_f; st->u = u; st->n_unclipped_wins = nwin;
  }
  orc_clipnorm_adam(c, s->params, s->grads, s->adam_m, s->adam_v, s->betap, eta, 0.5);
}

/* xoshiro256++ Fisher–Yates: what Random.shuffle costs the reference (serial, a few ns per element) */
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static void fast_shuffle(int32_t* perm, int32_t n, uint64_t seed) {
  uint64_t st[4] = {seed ^ 0x9E3779B97F4A7C15ull, seed * 0xBF58476D1CE4E5B9ull + 1, ~seed, 0x94D049BB133111EBull};
  for (int32_t i = n - 1; i >= 1; --i) {
    const uint64_t r = rotl(st[0] + st[3], 23) + st[0], t = st[1] << 17;
    st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3]; st[2] ^= t; st[3] = rotl(st[3], 45);
    const uint32_t j = (uint32_t)(((unsigned __int128)r * (uint64_t)(i + 1)) >> 64);
    const int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
  }
}

/* One pass of the ppo.jl:117-253 loop body. gen_perm = 0: s->perm is used as it stands for every epoch (the cross-check against
 * orc_iterate); 1: a fresh serial Fisher–Yates per epoch like the reference. Returns 0, or 1 when the shape is not covered. */
int32_t orc_batched_iterate(const orc_config* c, orc_state* s, int32_t num_updates_total, int32_t gen_perm, orc_stats* stats) {
  if (c->hidden != HMAX || c->obs_dim > 8 || c->n_act > 4 || c->env_kind != 0 || c->gae_mode != 0) return 1;
  double eta = (double)c->lr;
  if (c->anneal_lr) eta = (1.0 - ((double)(s->iteration + 1) - 1.0) / (double)num_updates_total) * (double)c->lr;
  batched_rollout(c, s);
  const int nt = c->num_envs, k = c->num_steps, B = nt * k, M = B / c->num_minibatches;
  float* nv = (float*)calloc((size_t)nt, sizeof(float));   /* compat mode never reads the bootstrap (Q10) */
  orc_gae_batch(s->value, s->reward, s->terminal, nv, s->next_done, nt, k, c->gamma, c->gae_lambda, c->gae_mode, s->adv, s->ret);
  free(nv);
  if (gen_perm) for (int i = 0; i < B; ++i) s->perm[i] = i;
  for (int ep = 0; ep < c->update_epochs; ++ep) {
    if (gen_perm) fast_shuffle(s->perm, B, c->seed + s->iteration * 131u + (uint64_t)ep);
    for (int mb = 0; mb < c->num_minibatches; ++mb)
      batched_update(c, s, s->perm + (size_t)mb * M, M, eta, stats ? stats + ep * c->num_minibatches + mb : NULL);
  }
  s->iteration += 1;
  return 0;
}
