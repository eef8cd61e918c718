// stats.hpp — the "Training Statistics" record (ppo.jl:243-247) from the four loss sums of one optimiser step.
#pragma once
#include "ppo_ctx.hpp"

namespace crl {

// "Training Statistics" (ppo.jl:247) from the (all-reduced) sums msg[P..P+3]; mode 0 also raises the value-loss
// speculation flag (u > 0), mode 1 is the re-evaluation after the exact critic pass.
__device__ __forceinline__ void compute_stats4(float s0, float s1, float s2, float s3, const DevCfg& c, double Mglobal, const double* adv_ms,
                                               int mb, double* vfix, crl_ppo_stats* out, int mode) {
  const double pg = (double)s0 / Mglobal;
  const double ent = (double)(float)((double)s1 / ((double)c.A * Mglobal));
  const double u = (double)(float)((double)s2 / Mglobal);
  const double vl = 0.5 * (double)(float)((double)s3 / Mglobal);
  if (mode == 0) {
    vfix[0] = u;
    vfix[3] = (c.clip_vloss && u > 0.0) ? 1.0 : 0.0;
    if (vfix[3] != 0.0) vfix[4] = 1.0;  // sticky: lets a data-parallel run fail loudly (no exact pass there yet)
    out->n_unclipped_wins = 0.0;
  } else {
    out->n_unclipped_wins = vfix[1];
  }
  out->pg_loss = pg; out->entropy_loss = ent; out->v_loss = vl; out->u_value = u;
  out->loss = pg - (double)(c.ent_coeff * (float)ent) + (double)c.v_coef * vl;
  out->adv_mean = (double)(float)adv_ms[2 * mb]; out->adv_std = (double)(float)adv_ms[2 * mb + 1];
}

__device__ __forceinline__ void compute_stats(const float* msg, int P, const DevCfg& c, double Mglobal, const double* adv_ms,
                                              int mb, double* vfix, crl_ppo_stats* out, int mode) {
  compute_stats4(msg[P], msg[P + 1], msg[P + 2], msg[P + 3], c, Mglobal, adv_ms, mb, vfix, out, mode);
}

struct StatsArgs {
  DevCfg c; double Mglobal; const double* adv_ms; int mb; double* vfix; crl_ppo_stats* out;
  int fused;  // 1: the last block of reduce_kernel also writes the statistics (single-GPU: sums are already global)
  float* dscale;   // [4]: G actor, G critic, then (as unsigned) the launch's largest |δ2| per role — fp16x2 weight gradient
};

}  // namespace crl
