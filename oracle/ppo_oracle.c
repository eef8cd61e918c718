/*
 * ppo_oracle.c — CPU restatement of sash-a/CleanRL.jl src/algorithms/ppo.jl (TEST INFRASTRUCTURE ONLY;
 * see ppo_oracle.h: PARITY UNPINNED). Every function cites the reference lines it follows; semantics of
 * un-vendored third-party packages are marked [3P-memory] (recalled from the pinned version's public
 * source, Manifest.toml) — see VERIFY_WITH_JULIA.md.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile). Contraction is off so every f32
 * operation rounds once as in Julia; fused multiply-adds appear only where Julia's evalpoly/muladd emits them.
 */
#include "ppo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ */
/* Parameter layout: Flux.params(actor, critic) (ppo.jl:196); networks.jl:36-49 builds                 */
/* actor = Chain(Dense(obs,h,tanh_fast), Dense(h,h,tanh_fast), Dense(h,A)), critic likewise with out 1 */
/* ------------------------------------------------------------------------------------------------ */
void orc_param_offsets(const orc_config* c, int32_t* o) {
  int h = c->hidden, d = c->obs_dim, A = c->n_act;
  int sizes[12] = {h * d, h, h * h, h, A * h, A, h * d, h, h * h, h, 1 * h, 1};
  o[0] = 0;
  for (int i = 0; i < 12; ++i) o[i + 1] = o[i] + sizes[i];
}
int32_t orc_param_count(const orc_config* c) {
  int32_t o[13];
  orc_param_offsets(c, o);
  return o[12];
}

/* NNlib 0.8.21 src/activations.jl tanh_fast(x::Float32) [3P-memory]: rational approximation,
 * evalpoly → muladd chains (fma on FMA hardware), cut-over to sign(x) at x^2 >= 66. networks.jl:6 */
float orc_tanh_fast(float x) {
  float x2 = x * x;
  float n = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 1.587199e-8f, 2.2332108e-5f), 0.0035974074f), 0.1346604f), 1.0f);
  float d = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 8.7767893e-7f, 0.0003453992f), 0.026262015f), 0.4679937f), 1.0f);
  if (x2 < 66.0f) return x * (n / d);
  return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : x);
}

/* Small-angle sin/cos shared bit-for-bit with the HIP env kernel (|x| <= ~0.3 rad in CartPole, where the
 * Taylor remainder is < 1e-11, i.e. below f32 rounding). Julia calls libm-quality sin/cos (<1 ulp);
 * this stays within 1 ulp of them on the env's range. */
float orc_sin_poly(float x) {
  float x2 = x * x;
  float p = fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f);
  return fmaf(x * x2, p, x);
}
float orc_cos_poly(float x) {
  float x2 = x * x;
  float p = fmaf(x2, fmaf(x2, fmaf(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f);
  return fmaf(x2, p, 1.0f);
}

/* ------------------------------------------------------------------------------------------------ */
/* Counter-based RNG. The reference draws from Julia's task-local Xoshiro (ppo.jl:26,82,194); no Julia */
/* stream can be matched here, so uniform draws are an INPUT of the restatement; this   */
/* Philox4x32-10 stream is the one both this oracle and the HIP path use for end-to-end runs.          */
/* ------------------------------------------------------------------------------------------------ */
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static void philox_env(uint64_t seed, uint32_t env_gid, uint64_t gstep, uint32_t stream, uint32_t out[4]) {
  orc_philox(env_gid, (uint32_t)gstep, (uint32_t)(gstep >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
/* rand(Float64)-shaped draw: 53 random bits * 2^-53 (Julia ≥1.7 Xoshiro float sampling shape) */
double orc_u53(uint64_t seed, uint32_t env_gid, uint64_t gstep, uint32_t stream) {
  uint32_t o[4];
  philox_env(seed, env_gid, gstep, stream, o);
  uint64_t bits = (((uint64_t)o[0] << 32) | o[1]) >> 11;
  return (double)bits * 0x1.0p-53;
}

/* ------------------------------------------------------------------------------------------------ */
/* MLP forward: Flux Dense = σ.(W*x .+ b) (networks.jl:10,42-46), f32 throughout (ppo.jl:87 Flux.f32)  */
/* ------------------------------------------------------------------------------------------------ */
static void dense(const float* W, const float* b, const float* x, int out, int in, float* y, int act) {
  for (int o = 0; o < out; ++o) {
    float acc = 0.0f;
    for (int i = 0; i < in; ++i) acc += W[o + out * i] * x[i];
    acc += b[o];
    y[o] = act ? orc_tanh_fast(acc) : acc;
  }
}
void orc_mlp_forward(const orc_config* c, const float* params, int net, const float* x, float* out, float* h1,
                     float* h2) {
  int32_t o[13];
  orc_param_offsets(c, o);
  int h = c->hidden, base = net ? 6 : 0, n_out = net ? 1 : c->n_act;
  float t1[1024], t2[1024];
  if (!h1) h1 = t1;
  if (!h2) h2 = t2;
  dense(params + o[base + 0], params + o[base + 1], x, h, c->obs_dim, h1, 1);
  dense(params + o[base + 2], params + o[base + 3], h1, h, h, h2, 1);
  dense(params + o[base + 4], params + o[base + 5], h2, n_out, h, out, 0);
}

/* NNlib 0.8.21 softmax / logsoftmax over dims=1 [3P-memory] (ppo.jl:23-24,36-37) */
static void softmax_col(const float* z, int A, float* p, float* lp) {
  float m = z[0];
  for (int a = 1; a < A; ++a) m = z[a] > m ? z[a] : m;
  float s = 0.0f;
  for (int a = 0; a < A; ++a) { p[a] = expf(z[a] - m); s += p[a]; }
  for (int a = 0; a < A; ++a) p[a] = p[a] / s;
  float ls = 0.0f;
  for (int a = 0; a < A; ++a) { lp[a] = z[a] - m; ls += expf(lp[a]); }
  float l = logf(ls);
  for (int a = 0; a < A; ++a) lp[a] = lp[a] - l;
}

/* StatsBase 0.33.21 sample(rng, wv::AbstractWeights) [3P-memory] (ppo.jl:26):
 * t = rand()*sum(wv); i=1; cw=wv[1]; while cw < t && i < n; i+=1; cw += wv[i]; end.
 * sum(wv) is the Float32 sum stored at Weights construction; cw is Float32, t Float64. */
static int sample_weights(const float* p, int A, double u, double* margin) {
  float sw = 0.0f;
  for (int a = 0; a < A; ++a) sw += p[a];
  double t = u * (double)sw;
  int i = 0;
  float cw = p[0];
  double mg = fabs((double)cw - t);
  while ((double)cw < t && i < A - 1) {
    i += 1;
    cw += p[i];
    if (i < A - 1) { double d = fabs((double)cw - t); if (d < mg) mg = d; }
  }
  if (margin) {
    /* distance to every interior knot, relative to the total weight */
    float cc = 0.0f; double best = 1e300;
    for (int a = 0; a < A - 1; ++a) { cc += p[a]; double d = fabs((double)cc - t); if (d < best) best = d; }
    *margin = best / (double)sw;
  }
  (void)mg;
  return i;
}

/* OpenMP team size for a loop of n items: one thread per `grain` items, at most the machine's. The parity tests call these
 * functions on a few hundred samples from a host with hundreds of hardware threads, possibly shared with other jobs: a full team
 * per call (and, in orc_loss_grad, a P-sized accumulator per thread) cost more than the loop. */
static int orc_team(long n, long grain) {
  int t = 1;
#ifdef _OPENMP
  t = omp_get_max_threads();
#endif
  long want = n / grain;
  if (want < 1) want = 1;
  return want < t ? (int)want : t;
}

/* get_action (ppo.jl:21-32) + value = critic(next_obs) (ppo.jl:128) */
void orc_get_action(const orc_config* c, const float* params, const float* obs, const double* u, int32_t n,
                    int32_t* action, float* logprob, float* value, double* margin) {
  int A = c->n_act, d = c->obs_dim;
#pragma omp parallel for schedule(static) num_threads(orc_team(n, 64))
  for (int b = 0; b < n; ++b) {
    float z[16], p[16], lp[16], v;
    orc_mlp_forward(c, params, 0, obs + (size_t)d * b, z, NULL, NULL);
    softmax_col(z, A, p, lp);
    int a = sample_weights(p, A, u[b], margin ? margin + b : NULL);
    action[b] = a;
    logprob[b] = lp[a];
    if (value) { orc_mlp_forward(c, params, 1, obs + (size_t)d * b, &v, NULL, NULL); value[b] = v; }
  }
}

/* logprob_actions (ppo.jl:34-45); entropy = -sum.(probs .* logprobs) is element-wise (Q3) */
void orc_logprob_actions(const orc_config* c, const float* params, const float* obs, const int32_t* actions,
                         int32_t n, float* logprob, float* entropy) {
  int A = c->n_act, d = c->obs_dim;
#pragma omp parallel for schedule(static) num_threads(orc_team(n, 64))
  for (int b = 0; b < n; ++b) {
    float z[16], p[16], lp[16];
    orc_mlp_forward(c, params, 0, obs + (size_t)d * b, z, NULL, NULL);
    softmax_col(z, A, p, lp);
    logprob[b] = lp[actions[b]];
    for (int a = 0; a < A; ++a) entropy[a + (size_t)A * b] = -(p[a] * lp[a]);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* gae (ppo.jl:48-73). nonterm = 1.0 .- terminals is Float64 (ppo.jl:63), gae = 0.0 Float64 (:65);     */
/* δ = r[t] + γ*nonterm[t+1]*v[t+1] - v[t] evaluates left to right in Float64 from the first product;   */
/* γ*λ is a Float32 product (both T) before it meets nonterm. Stored Float32 (similar(rewards), :62).   */
/* The loop runs t = k-1:-1:1 (:66) so slot k is never written (Q1): compat defines it as 0.            */
/* ------------------------------------------------------------------------------------------------ */
void orc_gae(const float* values, ptrdiff_t vs, const float* rewards, ptrdiff_t rs, const uint8_t* terminals,
             ptrdiff_t ts, int32_t k, float gamma, float lambda, int32_t mode, float* adv, ptrdiff_t as) {
  double gae = 0.0;
  float gl = gamma * lambda; /* Float32 product */
  int t_hi = mode ? k - 1 : k - 2;
  if (!mode) adv[(ptrdiff_t)(k - 1) * as] = 0.0f;
  for (int t = t_hi; t >= 0; --t) {
    double nonterm = 1.0 - (double)(terminals[(ptrdiff_t)(t + 1) * ts] ? 1 : 0);
    double delta = (double)rewards[(ptrdiff_t)t * rs] + ((double)gamma * nonterm) * (double)values[(ptrdiff_t)(t + 1) * vs] -
                   (double)values[(ptrdiff_t)t * vs];
    gae = delta + (((double)gl * nonterm) * gae);
    adv[(ptrdiff_t)t * as] = (float)gae;
  }
}

/* ppo.jl:173-181: per-env rows of hcat(value, next_values') etc.; returns = advantages + value */
void orc_gae_batch(const float* value, const float* reward, const uint8_t* terminal, const float* next_value,
                   const uint8_t* next_done, int32_t nt, int32_t k, float gamma, float lambda, int32_t mode,
                   float* adv, float* ret) {
#pragma omp parallel for schedule(static) num_threads(orc_team(nt, 16))
  for (int e = 0; e < nt; ++e) {
    float vrow[4097]; uint8_t trow[4097];
    float* vr = k + 1 <= 4097 ? vrow : (float*)malloc(sizeof(float) * (k + 1));
    uint8_t* tr = k + 1 <= 4097 ? trow : (uint8_t*)malloc(k + 1);
    for (int t = 0; t < k; ++t) { vr[t] = value[e + (size_t)nt * t]; tr[t] = terminal[e + (size_t)nt * t]; }
    vr[k] = next_value ? next_value[e] : 0.0f;
    tr[k] = next_done ? next_done[e] : 0;
    orc_gae(vr, 1, reward + e, nt, tr, 1, k, gamma, lambda, mode, adv + e, nt);
    for (int t = 0; t < k; ++t) ret[e + (size_t)nt * t] = adv[e + (size_t)nt * t] + value[e + (size_t)nt * t];
    if (vr != vrow) free(vr);
    if (tr != trow) free(tr);
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Loss closure + analytic backward (ppo.jl:202-244)                           */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float h1[1024], h2[1024]; } acts_t;

static void backprop_net(const orc_config* c, const float* params, int net, const int32_t* o, const float* x,
                         const acts_t* ac, const float* dout, double* g) {
  /* standard Dense pullbacks; tanh_fast pullback = 1 - y^2 (NNlib scalar rule [3P-memory]) */
  int h = c->hidden, d = c->obs_dim, base = net ? 6 : 0, n_out = net ? 1 : c->n_act;
  const float* W3 = params + o[base + 4];
  const float* W2 = params + o[base + 2];
  float d2[1024], d1[1024];
  for (int i = 0; i < h; ++i) {
    float s = 0.0f;
    for (int a = 0; a < n_out; ++a) s += W3[a + n_out * i] * dout[a];
    d2[i] = s * (1.0f - ac->h2[i] * ac->h2[i]);
  }
  for (int a = 0; a < n_out; ++a) {
    g[o[base + 5] + a] += dout[a];
    for (int i = 0; i < h; ++i) g[o[base + 4] + a + n_out * i] += (double)dout[a] * ac->h2[i];
  }
  for (int i = 0; i < h; ++i) {
    float s = 0.0f;
    for (int j = 0; j < h; ++j) s += W2[j + h * i] * d2[j];
    d1[i] = s * (1.0f - ac->h1[i] * ac->h1[i]);
  }
  for (int j = 0; j < h; ++j) {
    g[o[base + 3] + j] += d2[j];
    for (int i = 0; i < h; ++i) g[o[base + 2] + j + h * i] += (double)d2[j] * ac->h1[i];
  }
  for (int j = 0; j < h; ++j) {
    g[o[base + 1] + j] += d1[j];
    for (int i = 0; i < d; ++i) g[o[base + 0] + j + h * i] += (double)d1[j] * x[i];
  }
}

void orc_loss_grad(const orc_config* c, const float* params, const float* states, const int32_t* actions,
                   const float* logprobs, const float* values, const float* advantages, const float* returns,
                   const int32_t* mb_inds, int32_t M, const double* adv_stats, float* grads, orc_stats* st) {
  int32_t o[13];
  orc_param_offsets(c, o);
  const int P = o[12], A = c->n_act, d = c->obs_dim;
  const float eps = c->clip_coef;
  const float lo = 1 - eps, hi = 1 + eps; /* ppo.jl:227: Int ± Float32 → Float32 */

  /* advantage statistics (ppo.jl:221): mean/std of the Float32 view → Float32 (std corrected, n-1) */
  double mean_d = 0.0, var_d = 0.0;
  if (adv_stats) { mean_d = adv_stats[0]; var_d = adv_stats[1] * adv_stats[1]; }
  else {
    for (int j = 0; j < M; ++j) mean_d += advantages[mb_inds[j]];
    mean_d /= M;
    for (int j = 0; j < M; ++j) { double t = advantages[mb_inds[j]] - mean_d; var_d += t * t; }
    var_d /= (M - 1);
  }
  const float mean_f = (float)mean_d;
  const float std_f = adv_stats ? (float)adv_stats[1] : (float)sqrt(var_d);
  const double denom = (double)std_f + 1e-8; /* Float64 literal promotes (Q5) */

  /* pass 1: newvalue for the whole minibatch → u = mean(newvalue .- mb_returns .^ 2) (ppo.jl:232, Q4) */
  float* newv = (float*)malloc(sizeof(float) * (size_t)M);
  double usum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : usum) num_threads(orc_team(M, 64))
  for (int j = 0; j < M; ++j) {
    int s = mb_inds[j];
    float v;
    orc_mlp_forward(c, params, 1, states + (size_t)d * s, &v, NULL, NULL);
    newv[j] = v;
    float R = returns[s];
    usum += (double)(v - R * R);
  }
  const float u = (float)(usum / M);
  double nwin = 0.0;
  if (c->clip_value_loss) {
    for (int j = 0; j < M; ++j) {
      int s = mb_inds[j];
      float dv = newv[j] - values[s];
      float cl = dv < -eps ? -eps : (dv > eps ? eps : dv);
      float vc = values[s] + cl;
      float q = (vc - returns[s]) * (vc - returns[s]);
      if (u > q) nwin += 1.0;
    }
  }

  const int nthreads = orc_team(M, 64);
  double* gacc = (double*)calloc((size_t)P * nthreads, sizeof(double));
  double pg_sum = 0.0, vmax_sum = 0.0, ent_sum = 0.0;

#pragma omp parallel reduction(+ : pg_sum, vmax_sum, ent_sum) num_threads(nthreads)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double* g = gacc + (size_t)P * tid;
#pragma omp for schedule(static)
    for (int j = 0; j < M; ++j) {
      int s = mb_inds[j];
      const float* x = states + (size_t)d * s;
      acts_t aa, ac;
      float z[16], p[16], lp[16], v;
      orc_mlp_forward(c, params, 0, x, z, aa.h1, aa.h2);
      orc_mlp_forward(c, params, 1, x, &v, ac.h1, ac.h2);
      softmax_col(z, A, p, lp);
      int a = actions[s];
      float nlp = lp[a];
      double H = 0.0;
      for (int i = 0; i < A; ++i) { float e = -(p[i] * lp[i]); ent_sum += e; H += e; }

      /* policy loss (ppo.jl:219-228) */
      double Ahat = (double)(advantages[s] - mean_f) / denom;
      float logratio = nlp - logprobs[s];
      float ratio = expf(logratio);
      float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
      double pg1 = -Ahat * (double)ratio, pg2 = -Ahat * (double)rc;
      double dnlp;
      if (pg1 > pg2) { pg_sum += pg1; dnlp = pg1; } /* d(-Â ρ)/dnlp = -Â ρ */
      else { pg_sum += pg2; dnlp = (ratio >= lo && ratio <= hi) ? pg1 : 0.0; }
      dnlp /= M;

      /* value loss (ppo.jl:231-240) */
      double dv;
      float R = returns[s];
      if (c->clip_value_loss) {
        float dvv = v - values[s];
        float cl = dvv < -eps ? -eps : (dvv > eps ? eps : dvv);
        float vc = values[s] + cl;
        float q = (vc - R) * (vc - R);
        int q_wins = !(u > q); /* max.(u, q): ties → second argument */
        vmax_sum += q_wins ? (double)q : (double)u;
        double inner = (q_wins && dvv >= -eps && dvv <= eps) ? 2.0 * (double)(vc - R) : 0.0;
        dv = (double)c->v_coef * 0.5 / M * (nwin / M + inner);
      } else {
        float e = v - R;
        vmax_sum += (double)(e * e);
        dv = (double)c->v_coef * 0.5 / M * 2.0 * (double)e;
      }

      float dz[16];
      for (int i = 0; i < A; ++i) {
        double t = dnlp * ((i == a ? 1.0 : 0.0) - (double)p[i]) +
                   (double)c->ent_coeff / ((double)A * M) * (double)p[i] * ((double)lp[i] + H);
        dz[i] = (float)t;
      }
      float dvf = (float)dv;
      backprop_net(c, params, 0, o, x, &aa, dz, g);
      backprop_net(c, params, 1, o, x, &ac, &dvf, g);
    }
  }
  for (int i = 0; i < P; ++i) {
    double s = 0.0;
    for (int t = 0; t < nthreads; ++t) s += gacc[(size_t)P * t + i];
    grads[i] = (float)s;
  }
  free(gacc);
  free(newv);
  if (st) {
    st->pg_loss = pg_sum / M;
    st->v_loss = 0.5 * (double)(float)(vmax_sum / M); /* 0.5 * mean(Float32 array) */
    st->entropy_loss = (double)(float)(ent_sum / ((double)A * M));
    st->loss = st->pg_loss - (double)(c->ent_coeff * (float)st->entropy_loss) + (double)c->v_coef * st->v_loss;
    st->adv_mean = mean_f; st->adv_std = std_f; st->u = u; st->n_unclipped_wins = nwin;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* Flux 0.13.4 Optimiser(ClipNorm(0.5), Adam(η)) [3P-memory] (ppo.jl:93,250), applied per array (Q9):  */
/*  ClipNorm: n = norm(Δ); n > thresh && rmul!(Δ, thresh/n)                                            */
/*  Adam: mt = β1*mt + (1-β1)*Δ ; vt = β2*vt + (1-β2)*Δ^2 ; Δ = mt/(1-β1p) / (√(vt/(1-β2p)) + ϵ) * η ;    */
/*        βp .*= β ; x .-= Δ.   β=(0.9,0.999), ϵ=1e-8, scalars Float64, arrays Float32.                 */
/* ------------------------------------------------------------------------------------------------ */
void orc_clipnorm_adam(const orc_config* c, float* params, float* grads, float* m, float* v, double* betap,
                       double eta, double clip_thresh) {
  int32_t o[13];
  orc_param_offsets(c, o);
  const double b1 = 0.9, b2 = 0.999, epsn = 1e-8;
  for (int a = 0; a < 12; ++a) {
    double ss = 0.0;
    for (int i = o[a]; i < o[a + 1]; ++i) ss += (double)grads[i] * grads[i];
    float nrm = (float)sqrt(ss);
    if ((double)nrm > clip_thresh) {
      double sc = clip_thresh / (double)nrm;
      for (int i = o[a]; i < o[a + 1]; ++i) grads[i] = (float)((double)grads[i] * sc);
    }
    double* bp = betap + 2 * a;
    for (int i = o[a]; i < o[a + 1]; ++i) {
      double g = grads[i];
      m[i] = (float)(b1 * (double)m[i] + (1 - b1) * g);
      v[i] = (float)(b2 * (double)v[i] + (1 - b2) * g * g);
      double delta = (double)m[i] / (1 - bp[0]) / (sqrt((double)v[i] / (1 - bp[1])) + epsn) * eta;
      float df = (float)delta;
      params[i] = params[i] - df;
    }
    bp[0] *= b1; bp[1] *= b2;
  }
}

/* ------------------------------------------------------------------------------------------------ */
/* CartPoleEnv{Float32} step — ReinforcementLearningEnvironments 0.6.12 [3P-memory] (ppo.jl:82,130):   */
/* Euler update; the `4 / 3` literal is Float64, so thetaacc/xacc and the two velocity updates promote   */
/* to Float64 before being stored back into the Float32 state. reward = done ? 0 : 1 (ppo.jl:132, Q12). */
/* ------------------------------------------------------------------------------------------------ */
void orc_cartpole_step(float* s, int32_t* t, int32_t action, int32_t max_steps, int32_t* done) {
  const float gravity = 9.8f, masspole = 0.1f, totalmass = 1.1f, halflength = 0.5f, pml = 0.05f;
  const float forcemag = 10.0f, dt = 0.02f, ththr = 0.20943951f, xthr = 2.4f;
  *t += 1;
  float force = action == 1 ? forcemag : -forcemag; /* Julia a == 2 */
  float x = s[0], xdot = s[1], theta = s[2], thetadot = s[3];
  (void)x;
  float costheta = orc_cos_poly(theta), sintheta = orc_sin_poly(theta);
  float tmp = (force + (pml * (thetadot * thetadot)) * sintheta) / totalmass;
  float num = gravity * sintheta - costheta * tmp;
  double den = (double)halflength * (4.0 / 3.0 - (double)((masspole * (costheta * costheta)) / totalmass));
  double thetaacc = (double)num / den;
  double xacc = (double)tmp - (((double)pml * thetaacc) * (double)costheta) / (double)totalmass;
  s[0] = s[0] + dt * xdot;
  s[1] = (float)((double)s[1] + (double)dt * xacc);
  s[2] = s[2] + dt * thetadot;
  s[3] = (float)((double)s[3] + (double)dt * thetaacc);
  *done = (fabsf(s[0]) > xthr) || (fabsf(s[2]) > ththr) || (*t > max_steps);
}

/* The same step with libm's sinf / cosf — what Julia's sin / cos (< 1 ulp) amount to. orc_cartpole_step above replaces them by the
 * small-angle polynomials the HIP env kernel evaluates (csrc/env.hpp), so that CPU and GPU trajectories can be compared BIT FOR BIT:
 * that polynomial ORIGINATES IN THE KERNEL, not in the reference. This variant is the reference-side statement; the
 * tests bound the distance of both the polynomial oracle and the HIP rollout from it, step by step from identical states
 * (tests/test_oracle.py::test_cartpole_polynomial_step_tracks_libm_step, tests/test_gpu_parity.py::test_rollout_steps_track_the_libm_cartpole). */
void orc_cartpole_step_libm(float* s, int32_t* t, int32_t action, int32_t max_steps, int32_t* done) {
  const float gravity = 9.8f, masspole = 0.1f, totalmass = 1.1f, halflength = 0.5f, pml = 0.05f;
  const float forcemag = 10.0f, dt = 0.02f, ththr = 0.20943951f, xthr = 2.4f;
  *t += 1;
  float force = action == 1 ? forcemag : -forcemag;
  float xdot = s[1], theta = s[2], thetadot = s[3];
  float costheta = cosf(theta), sintheta = sinf(theta);
  float tmp = (force + (pml * (thetadot * thetadot)) * sintheta) / totalmass;
  float num = gravity * sintheta - costheta * tmp;
  double den = (double)halflength * (4.0 / 3.0 - (double)((masspole * (costheta * costheta)) / totalmass));
  double thetaacc = (double)num / den;
  double xacc = (double)tmp - (((double)pml * thetaacc) * (double)costheta) / (double)totalmass;
  s[0] = s[0] + dt * xdot;
  s[1] = (float)((double)s[1] + (double)dt * xacc);
  s[2] = s[2] + dt * thetadot;
  s[3] = (float)((double)s[3] + (double)dt * thetaacc);
  *done = (fabsf(s[0]) > xthr) || (fabsf(s[2]) > ththr) || (*t > max_steps);
}
/* n independent single steps from given states (4, n) with given actions: libm = 1 → the sinf/cosf variant, 0 → the polynomial one */
void orc_cartpole_step_batch(const float* s_in, const int32_t* action, int32_t n, int32_t libm, float* s_out, uint8_t* done_out) {
  for (int i = 0; i < n; ++i) {
    float s[4]; int32_t t = 0, dn = 0;
    memcpy(s, s_in + 4 * (size_t)i, sizeof(s));
    if (libm) orc_cartpole_step_libm(s, &t, action[i], 500, &dn); else orc_cartpole_step(s, &t, action[i], 500, &dn);
    memcpy(s_out + 4 * (size_t)i, s, sizeof(s));
    if (done_out) done_out[i] = (uint8_t)dn;
  }
}

/* reset!: state = T(0.1) * rand(rng, T, 4) .- T(0.05) [3P-memory]; rand(Float32) shape = 24 bits * 2^-24 */
void orc_env_reset(const orc_config* c, float* s, uint32_t env_gid, uint64_t gstep, uint32_t stream) {
  uint32_t o[4];
  philox_env(c->seed, env_gid, gstep, stream, o);
  for (int i = 0; i < 4; ++i) {
    float r = (float)(o[i] >> 8) * 0x1.0p-24f;
    s[i] = 0.1f * r - 0.05f;
  }
}

/* synthetic obs8/act4 env (BASELINE config C3; the reference has no LunarLander): obs ~ U(-1,1)^d,
 * reward ~ U(-1,1), done ~ Bernoulli(1/200); all from the env's Philox stream. */
static void synth_step(const orc_config* c, float* s, uint32_t env_gid, uint64_t gstep, float* reward, int32_t* done) {
  uint32_t o[4];
  int d = c->obs_dim;
  for (int q = 0; q < (d + 3) / 4; ++q) {
    philox_env(c->seed, env_gid, gstep, 8 + q, o);
    for (int i = 0; i < 4 && 4 * q + i < d; ++i) s[4 * q + i] = (float)(o[i] >> 8) * 0x1.0p-23f - 1.0f;
  }
  philox_env(c->seed, env_gid, gstep, 3, o);
  *reward = (float)(o[0] >> 8) * 0x1.0p-23f - 1.0f;
  *done = (o[1] % 200u) == 0u;
}

/* ------------------------------------------------------------------------------------------------ */
/* Loop state                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
orc_state* orc_state_create(const orc_config* c) {
  orc_state* s = (orc_state*)calloc(1, sizeof(orc_state));
  size_t nt = c->num_envs, k = c->num_steps, B = nt * k, d = c->obs_dim, P = orc_param_count(c);
  s->obs = (float*)calloc(B * d, 4); s->action = (int32_t*)calloc(B, 4); s->logprob = (float*)calloc(B, 4);
  s->reward = (float*)calloc(B, 4); s->terminal = (uint8_t*)calloc(B, 1); s->value = (float*)calloc(B, 4);
  s->adv = (float*)calloc(B, 4); s->ret = (float*)calloc(B, 4);
  s->env_state = (float*)calloc(nt * d, 4); s->env_t = (int32_t*)calloc(nt, 4); s->cur_obs = (float*)calloc(nt * d, 4);
  s->next_done = (uint8_t*)calloc(nt, 1); s->ep_return = (float*)calloc(nt, 4); s->ep_length = (int32_t*)calloc(nt, 4);
  s->params = (float*)calloc(P, 4); s->grads = (float*)calloc(P, 4); s->adam_m = (float*)calloc(P, 4);
  s->adam_v = (float*)calloc(P, 4); s->perm = (int32_t*)calloc(B, 4);
  for (int a = 0; a < 12; ++a) { s->betap[2 * a] = 0.9; s->betap[2 * a + 1] = 0.999; }
  for (size_t i = 0; i < B; ++i) s->perm[i] = (int32_t)i;
  return s;
}
void orc_state_destroy(orc_state* s) {
  if (!s) return;
  free(s->obs); free(s->action); free(s->logprob); free(s->reward); free(s->terminal); free(s->value);
  free(s->adv); free(s->ret); free(s->env_state); free(s->env_t); free(s->cur_obs); free(s->next_done);
  free(s->ep_return); free(s->ep_length); free(s->params); free(s->grads); free(s->adam_m); free(s->adam_v);
  free(s->perm); free(s);
}

/* ppo.jl:80-83,112-115: envs are reset at construction; next_obs = state(env); next_done = false */
void orc_env_init(const orc_config* c, orc_state* s) {
  int d = c->obs_dim;
  for (int e = 0; e < c->num_envs; ++e) {
    uint32_t gid = (uint32_t)(c->env_id_offset + e);
    if (c->env_kind == 0) orc_env_reset(c, s->env_state + (size_t)d * e, gid, 0, 2);
    else { float r; int32_t dn; synth_step(c, s->env_state + (size_t)d * e, gid, ~(uint64_t)0, &r, &dn); }
    s->env_t[e] = 0;
    memcpy(s->cur_obs + (size_t)d * e, s->env_state + (size_t)d * e, sizeof(float) * d);
    s->next_done[e] = 0; s->ep_return[e] = 0.0f; s->ep_length[e] = 0;
  }
  s->iteration = 0;
}

/* ppo.jl:123-166 */
void orc_rollout(const orc_config* c, orc_state* s) {
  const int nt = c->num_envs, k = c->num_steps, d = c->obs_dim, A = c->n_act;
  double epc = 0, eprs = 0, epls = 0;
#pragma omp parallel for schedule(static) reduction(+ : epc, eprs, epls) num_threads(orc_team(nt, 4))
  for (int e = 0; e < nt; ++e) {
    uint32_t gid = (uint32_t)(c->env_id_offset + e);
    float* es = s->env_state + (size_t)d * e;
    float* co = s->cur_obs + (size_t)d * e;
    for (int t = 0; t < k; ++t) {
      uint64_t gstep = s->iteration * (uint64_t)k + (uint64_t)t;
      size_t b = (size_t)e + (size_t)nt * t;
      s->ep_length[e] += 1;                                           /* ppo.jl:125 */
      float z[16], p[16], lp[16], v;
      orc_mlp_forward(c, s->params, 0, co, z, NULL, NULL);            /* ppo.jl:127 */
      softmax_col(z, A, p, lp);
      double u = orc_u53(c->seed, gid, gstep, 0);
      int a = sample_weights(p, A, u, NULL);
      orc_mlp_forward(c, s->params, 1, co, &v, NULL, NULL);           /* ppo.jl:128 */
      int32_t done; float rew;
      if (c->env_kind == 0) {
        orc_cartpole_step(es, &s->env_t[e], a, 500, &done);           /* ppo.jl:130 */
        rew = done ? 0.0f : 1.0f;                                     /* ppo.jl:132 */
      } else synth_step(c, es, gid, gstep, &rew, &done);
      memcpy(s->obs + b * d, co, sizeof(float) * d);                  /* ppo.jl:133-140 */
      s->action[b] = a; s->logprob[b] = lp[a]; s->reward[b] = rew;
      s->terminal[b] = s->next_done[e]; s->value[b] = v;
      memcpy(co, es, sizeof(float) * d);                              /* ppo.jl:143 (before reset!, Q7) */
      s->next_done[e] = (uint8_t)done;                                /* ppo.jl:144 */
      s->ep_return[e] += rew;                                         /* ppo.jl:145 */
      if (done) {                                                     /* ppo.jl:147-165 */
        epc += 1; eprs += s->ep_return[e]; epls += s->ep_length[e];
        s->ep_return[e] = 0.0f; s->ep_length[e] = 0;
        if (c->env_kind == 0) {
          orc_env_reset(c, es, gid, gstep, 1); s->env_t[e] = 0;        /* ppo.jl:164 */
          if (!c->stale_obs) memcpy(co, es, sizeof(float) * d);
        }
      }
    }
  }
  s->ep_count = epc; s->ep_return_sum = eprs; s->ep_length_sum = epls;
}

/* ppo.jl:169-181; next_values = critic(state(env)) is live only in fixed mode (Q10) */
void orc_compute_gae(const orc_config* c, orc_state* s) {
  const int nt = c->num_envs, k = c->num_steps, d = c->obs_dim;
  float* nv = (float*)malloc(sizeof(float) * nt);
  for (int e = 0; e < nt; ++e) orc_mlp_forward(c, s->params, 1, s->cur_obs + (size_t)d * e, nv + e, NULL, NULL);
  orc_gae_batch(s->value, s->reward, s->terminal, nv, s->next_done, nt, k, c->gamma, c->gae_lambda, c->gae_mode,
                s->adv, s->ret);
  free(nv);
}

/* ppo.jl:194 shuffle = Fisher–Yates (Random.shuffle! [3P-memory]: for i = n:-1:2, j = rand(1:i), swap).
 * Draws come from this build's Philox stream: j = floor(u64 * i / 2^64) over ctr=(i, epoch). */
void orc_shuffle_fy(int32_t* perm, int32_t n, uint64_t seed, uint64_t epoch_id) {
  for (int32_t i = n - 1; i >= 1; --i) {
    uint32_t o[4];
    orc_philox((uint32_t)i, (uint32_t)epoch_id, (uint32_t)(epoch_id >> 32), 0x5FFu, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    uint64_t r = ((uint64_t)o[0] << 32) | o[1];
    uint32_t j = (uint32_t)(((unsigned __int128)r * (uint64_t)(i + 1)) >> 64);
    int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp;
  }
}

/* Blocked Fisher–Yates (CRL_SHUFFLE_BLOCKED_FY): Rao–Sandelius split into sub-buckets of ~16 elements by two random
 * digits per element, exact Fisher–Yates inside every sub-bucket, sub-buckets concatenated in id order. A uniform draw
 * from S_n like ppo.jl:194's shuffle, but parallel; this is the build's own algorithm, restated here so the HIP kernels
 * can be checked bit for bit. Always starts from the identity (ppo.jl:191 b_inds = 1:batch_size). */
void orc_shuffle_blocked_fy(int32_t* perm, int32_t n, uint64_t seed, uint64_t epoch_id) {
  uint32_t K1 = 1;
  while ((uint64_t)K1 * 4096u < (uint64_t)n) K1 *= 2;
  const uint32_t G = K1 * 256u;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), e0 = (uint32_t)epoch_id, e1 = (uint32_t)(epoch_id >> 32);
  uint32_t* gid = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
  uint32_t* off = (uint32_t*)calloc((size_t)G + 1, sizeof(uint32_t));
  for (int32_t i = 0; i < n; ++i) {
    uint32_t o[4];
    orc_philox((uint32_t)i, e0, e1, 0xB0Cu, k0, k1, o);
    gid[i] = (o[0] & (K1 - 1)) * 256u + (o[1] & 255u);
    off[gid[i] + 1] += 1;
  }
  for (uint32_t g = 0; g < G; ++g) off[g + 1] += off[g];
  uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)G);
  memcpy(cur, off, sizeof(uint32_t) * (size_t)G);
  for (int32_t i = 0; i < n; ++i) perm[cur[gid[i]]++] = i; /* ascending i inside each sub-bucket */
  for (uint32_t g = 0; g < G; ++g) {
    int32_t* m = perm + off[g];
    const int32_t cnt = (int32_t)(off[g + 1] - off[g]);
    for (int32_t j = cnt - 1; j >= 1; --j) {
      uint32_t o[4];
      orc_philox(g, (uint32_t)j, e0 ^ 0x9E3779B9u, e1 ^ 0xF15A7E5u, k0, k1, o);
      const uint64_t r = ((uint64_t)o[0] << 32) | o[1];
      const uint32_t t = (uint32_t)(((unsigned __int128)r * (uint64_t)(j + 1)) >> 64);
      const int32_t tmp = m[j]; m[j] = m[t]; m[t] = tmp;
    }
  }
  free(gid); free(off); free(cur);
}

void orc_update_minibatch(const orc_config* c, orc_state* s, int32_t mb, double eta, orc_stats* st) {
  int B = c->num_envs * c->num_steps, M = B / c->num_minibatches;
  orc_loss_grad(c, s->params, s->obs, s->action, s->logprob, s->value, s->adv, s->ret, s->perm + (size_t)mb * M, M,
                NULL, s->grads, st);
  orc_clipnorm_adam(c, s->params, s->grads, s->adam_m, s->adam_v, s->betap, eta, 0.5);
}

/* One pass of the ppo.jl:117-253 loop body */
void orc_iterate(const orc_config* c, orc_state* s, int32_t num_updates_total, int32_t gen_perm, orc_stats* stats) {
  double eta = (double)c->lr;
  if (c->anneal_lr) { /* ppo.jl:118-121; update is 1-based */
    double frac = 1.0 - ((double)(s->iteration + 1) - 1.0) / (double)num_updates_total;
    eta = frac * (double)c->lr;
  }
  orc_rollout(c, s);
  orc_compute_gae(c, s);
  int B = c->num_envs * c->num_steps;
  if (gen_perm) for (int i = 0; i < B; ++i) s->perm[i] = i; /* ppo.jl:191 b_inds = 1:batch_size */
  for (int ep = 0; ep < c->update_epochs; ++ep) {
    if (gen_perm) orc_shuffle_fy(s->perm, B, c->seed, s->iteration * (uint64_t)c->update_epochs + ep);
    for (int mb = 0; mb < c->num_minibatches; ++mb)
      orc_update_minibatch(c, s, mb, eta, stats ? stats + ep * c->num_minibatches + mb : NULL);
  }
  s->iteration += 1;
}
