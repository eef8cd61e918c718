// update_args.hpp — launch arguments shared by the update kernels (update.hip).
#pragma once
#include "ppo_ctx.hpp"

namespace crl {

struct UpdateArgs {
  DevCfg c;
  const float* params;
  const SampleRec* recs;  // the batch as 64-byte records, buffer order (records.hip)
  const int32_t* perm = nullptr;  // this minibatch's slice of b_inds (ppo.jl:203-204): sample pos reads recs[perm[pos]]
  const double* adv_ms;   // [nmb][2] mean, std of the (global) minibatch advantages
  const double* vfix;     // [8] u, #{u > q}, -, flag, sticky flag
  float* gpart; double* lpart; float* newv;
  double* range_err;      // vfix[5]: set when a hidden-layer weight does not fit the fp16x2 window (mlp_x2.hpp)
  const float* dscale;    // [2] fp16x2 weight-gradient scale G of this launch, actor / critic
  unsigned* dmax;         // [2] largest |δ2| seen by this launch (float bits), turned into the next G by reduce_kernel
  unsigned* dw_miss = nullptr;  // 16-sample-tile kernel (update16.hpp): set when a tile did not fit G or a weight left the fp16 window — update_repair_kernel then redoes the minibatch as bf16x3
  int mb, mode, gstride;
  int nblk[2];            // blocks working on the actor / the critic
  int pmax;               // capacity (blocks per role) of the partial buffers
  int stagger;            // x3 kernel: start delay of waves 4-7 (units of 1024 clocks)
  int xcd_align = 0;      // 1 (both block counts multiples of 8): tile t is worked on by blocks with index ≡ t (mod 8) in BOTH roles — same XCD, same L2
  int prio_mode = 0;      // small launches (< 16 tiles per wave, where the feedback rule is off): 0 = nothing, 1 = the feedback rule anyway, 2 = static priority 1 for waves 4-7, 3 = priority alternating tile by tile
  double Mglobal;         // minibatch size over all ranks (the 1/M of every mean)
};


}  // namespace crl
