// update.hip — one PPO optimiser step's loss + gradient (ppo.jl:202-244 and the Zygote pullbacks behind it).
//
// Work split: actor and critic are independent networks that meet only in the scalar loss, so blocks alternate roles
// (even = actor, odd = critic); each wave walks 32-sample tiles of the minibatch:
//   read the tile's 64-byte sample records (already in minibatch order: records.hip) → forward (MFMA, weights in LDS as A-fragments, activations chained in registers)
//   → per-sample loss terms and output cotangent → backward:
//       dh = Wᵀ·δ        MFMA, B operand = δ straight from its C-fragment registers
//       dW2ᵀ += h1·δ2ᵀ   MFMA with K = samples: both operands transposed through a wave-private LDS tile [64][36]
//       dW1, dW3, biases  "lane = row" VALU sums over the same transposed tiles
// dW accumulates in registers across all tiles of the wave; waves → block through LDS; blocks → a per-block partial in
// HBM; a second kernel sums the partials in fixed order (bitwise reproducible; no float atomics).
// 196 MFMAs of 32x32x2 per (tile, network): 192 of them are the three 64x64 GEMMs (fwd, dW, dX).
//
// Value loss (ppo.jl:231-237, Q4): max.(u, q_b) with the SCALAR u = mean(v - R²) needs u before any critic
// cotangent exists. q_b ≥ 0, so whenever u ≤ 0 every max picks q_b: the kernel speculates on that, and
// reduce_kernel raises a flag if u > 0; only then do vfix_count_kernel and a critic-only exact pass rerun.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "common.hpp"
#include "mlp_x2.hpp"
#include "mlp_x3.hpp"
#include "ppo_ctx.hpp"
#include "stats.hpp"
#include "update_args.hpp"

namespace crl {

#ifdef CRL_EXP_STAMPS
// diagnostic build only (scripts/build_variant.sh stamps -DCRL_EXP_STAMPS; never in the library): per-wave wall-clock stamps (100 MHz)
// at the phase boundaries of the update kernel and of reduce_optim_kernel, read back by scripts/stamps_probe.py
__device__ unsigned long long crl_dbg_stamps[512 * 8 * 8];
#define CRL_STAMP(slot) do { if ((threadIdx.x & 63) == 0) crl_dbg_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (slot)] = wall_clock64(); } while (0)
// phase stamps of ONE tile per wave (its 21st): low words of the 100 MHz clock, kept in scalar registers until the tile ends
__device__ unsigned crl_dbg_tstamps[512 * 8 * 16];
// (ten scalar registers for all ten stamps push the kernel into spills: a build takes five of them — CRL_TS_SET 0: points 0-4 and 9, 1: points 0 and 5-9)
#ifndef CRL_TS_SET
#define CRL_TS_SET 0
#endif
constexpr int crl_ts_slot(int k) { return k == 0 ? 0 : k == 9 ? 5 : (CRL_TS_SET == 0 ? (k <= 4 ? k : -1) : (k >= 5 ? k - 4 : -1)); }
#define CRL_TS(k) do { if constexpr (crl_ts_slot(k) >= 0) { if (ts_on) ts[crl_ts_slot(k)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CRL_STAMP(slot) do { } while (0)
#define CRL_TS(k) do { } while (0)
#endif
constexpr int TSTRIDE = 36;
// phase boundary: orders the wave's LDS traffic AND stops the scheduler from moving register-only work across it (hoisted
// loads of the next phase were the source of the spills)
#define CRL_PHASE() do { wave_lds_fence(); __builtin_amdgcn_sched_barrier(0); } while (0)  // floats per row of the transposed tile: 144 B keeps b128 reads aligned and conflict-free

// a wave-uniform value into scalar registers
__device__ __forceinline__ float sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double sgpr(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the same through inline asm: the compiler drops a __builtin readfirstlane of a value it can prove uniform and then keeps the
// VALU-computed Float64 in vector registers; an asm v_readfirstlane_b32 cannot be folded away, so the result really is scalar
__device__ __forceinline__ double sgpr_hard(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned vlo = (unsigned)u, vhi = (unsigned)(u >> 32);
  unsigned lo, hi;
  asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(lo), "=s"(hi) : "v"(vlo), "v"(vhi));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Per-sample inputs of one tile: two 16-byte quarters of the sample's record (36 B of it are the fields of SURVEY §8d)
template <int D>
struct Gathered {
  float x[D];
  float f0, f1;  // actor: old logprob, advantage ; critic: old value, return
  int act;
};

template <int D, int ROLE>
__device__ __forceinline__ void gather(const UpdateArgs& a, int pos, Gathered<D>& g) {
  static_assert(D == 4, "SampleRec carries a 4-float observation");
  const int idx = a.perm ? a.perm[pos] : pos;
  const f32x4* r = reinterpret_cast<const f32x4*>(a.recs + idx);
  const f32x4 xv = r[0], q = r[ROLE == 0 ? 1 : 2];
  g.x[0] = xv[0]; g.x[1] = xv[1]; g.x[2] = xv[2]; g.x[3] = xv[3];
  if (ROLE == 0) { g.act = __float_as_int(q[0]); g.f0 = q[1]; g.f1 = q[2]; }
  else { g.act = 0; g.f0 = q[0]; g.f1 = q[1]; }
}

// A C-fragment tile (lane = sample j, registers = rows) into the wave's transposed scratch T[row][TSTRIDE]: 32 stores per lane.
// (Measured and dropped: 16 ds_write2_b32 instead — −0.2 % on the kernel.)
__device__ __forceinline__ void store_transposed(float* T, const f32x16 (&v)[2], int j, int hf) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) T[(32 * mt + rowmap(r, hf)) * TSTRIDE + j] = v[mt][r];
}

// One role (actor or critic) = RW waves of the block: `smem` is the role's weight image, `scratch` the first of its
// RW wave-private tiles. All barriers are block-wide and both roles execute the same number of them.
constexpr int SCR_FLOATS = 64 * TSTRIDE + TILE * 4 + 2 * TILE;
// bf16x3 kernels: the "skinny" gradient sums (dW1, db1, db2, dW3, db3: 10 floats per lane for the actor) accumulate in a
// wave-private LDS strip (read at the start of the phase that produces a partial sum, written back at its end) instead of in
// registers that stay live across the whole loop — those ten registers were what the 256-register budget was short of: the
// kernel now has no scratch at all (18 spilled registers before)
constexpr int ACC_SLOTS = 4 + 2 + 2 * 2;
constexpr int SCR_FLOATS_X3 = SCR_FLOATS + ACC_SLOTS * 64;

// ABL: bit mask of phases to leave out — the timing experiments behind the per-phase costs in DESIGN.md §3 (results are garbage);
// 0 in every instantiation of the library (the experiment kernels and their launch switch were removed in round 3).
// X2 (with X3): the forward and backward-data products run as fp16x2 (mlp_x2.hpp: three MFMAs per product instead of six, h1
// carried as 2^14·h1), the weight-gradient product stays on bf16x3.
// Returns false — before any work, uniformly for the block — only in the fp16x2 flavour when a hidden-layer weight of this role does
// not fit the fp16 window (|w| >= 255, mlp_x2.hpp): the caller then runs the bf16x3 flavour of the same role on the same LDS.
//
// Record prefetch (fp16x2 main pass, `pfslots` != nullptr). A tile starts with two DEPENDENT global round trips — perm[pos], then the
// random 64-byte record — which a wave with one partner on its SIMD cannot hide (≈2 µs of an ≈8 µs tile parked on vmcnt). So the
// records of the NEXT tile are fetched while this one computes, at no register cost: one global_load_lds_dwordx4 (LDS-DMA, per-lane
// source address, lane-linear destination) drops the tile's 32 observation quarters (lanes 0-31) and 32 role quarters (lanes 32-63)
// into a 1 KB wave-private LDS slot; the permutation entries it needs were fetched a tile earlier the same way
// (global_load_lds_dword into a 256-byte slot). The tile top then is `s_waitcnt vmcnt(0)` on loads issued ≈8 µs ago plus three LDS reads.
#ifndef CRL_PF_X3
#define CRL_PF_X3 0   // the record prefetch in update_x3_kernel (option gemm = 1) as well: measured SLOWER there (0.727 vs 0.715 ms per launch, three rounds on one box)
#endif
#ifndef CRL_PF_ENABLED
#define CRL_PF_ENABLED 1
#endif
constexpr int PF_SLOT_FLOATS = 256 + 64;   // 64 lanes x 16 B of records + 64 lanes x 4 B of permutation entries
template <int D, int A, int ROLE, bool EXACT, bool X3, int RW, int ABL = 0, bool X2 = false>
__device__ __forceinline__ bool update_role(const UpdateArgs& a, const int rb, float* smem, float* scratch, float* pfslots = nullptr) {
  constexpr int NOUT = ROLE == 0 ? A : 1;
  static_assert(!X2 || X3, "the fp16x2 flavour keeps the bf16x3 weight-gradient path");
  using I = typename std::conditional<X2, NetImageX2<D, NOUT>,
                                      typename std::conditional<X3, NetImageX3<D, NOUT, true>, NetImage<D, NOUT, true>>::type>::type;
  using P = NetParams<D, NOUT>;
  constexpr int SCR = X3 ? SCR_FLOATS_X3 : SCR_FLOATS;
  static_assert(64 * TSTRIDE + TILE * D + A * TILE <= SCR_FLOATS, "scratch too small");
  static_assert(D + 2 + 2 * NOUT <= ACC_SLOTS, "accumulator strip too small");
  const DevCfg& c = a.c;
  // the wave index is uniform (told to the compiler: the tile loop and its branches become scalar control flow)
  const int tid = threadIdx.x & (64 * RW - 1), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, hf = lane >> 5;
  constexpr int NT = 64 * RW;  // threads of this role
  float* img0 = smem;
  float* T0 = scratch + wave * SCR;
  const float* p = a.params + (ROLE ? NetParams<D, A>::SIZE : 0);
  CRL_STAMP(0);
  if constexpr (X2) {
    if (!stage_net_x2<D, NOUT>(img0, p, tid, NT, reinterpret_cast<int*>(scratch + RW * SCR))) {
      if (tid == 0 && rb == 0) a.range_err[0] = 1.0;   // informational: the bf16x3 fallback ran (crl_ppo_get_option "gemm_fallback_seen")
      __syncthreads();                                  // every thread has read the flag before the image is restaged
      return false;
    }
  }
  else if (X3) stage_net_x3<D, NOUT, true>(img0, p, tid, NT);
  else stage_net<D, NOUT, true>(img0, p, tid, NT);
  __syncthreads();

  CRL_STAMP(1);
  f32x16 dW2t[2][2];  // dW2ᵀ accumulators: [mj = h1-row block][ni = δ2-row block]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) dW2t[x][y][r] = 0.0f;
  // skinny sums: slot k of this lane — dW1[0..D), db1, db2, dW3[0..NOUT), db3[0..NOUT)
  constexpr int K_B1 = D, K_B2 = D + 1, K_W3 = D + 2, K_B3 = D + 2 + NOUT, NACC = D + 2 + 2 * NOUT;
  float racc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) racc[i] = 0.0f;
  float* ACC = scratch + wave * SCR + SCR_FLOATS + (threadIdx.x & 63);
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) ACC[64 * i] = 0.0f;
  }
  constexpr bool LACC = X3;
  // read at the start of the phase that produces the partial sum, written back at its end (plain LDS read / write: the strip
  // is private to the lane; ds_add_f32 atomics here cost +24 % on the whole kernel)
  auto acc_begin = [&](int k) -> float { if constexpr (LACC) return ACC[64 * k]; else return racc[k]; };
  auto acc_end = [&](int k, float v) { if constexpr (LACC) ACC[64 * k] = v; else racc[k] = v; };
  double ls0 = 0.0, ls1 = 0.0;

  const int M = c.M;
  const int ntiles = (M + TILE - 1) / TILE;
  const int nwaves = a.nblk[ROLE] * RW;
  // role constants: wave-uniform, pinned to scalar registers (the values loaded from adv_ms / vfix otherwise sit in vector
  // registers for the whole loop — eight of the registers that used to be spilled)
  const double invM = X2 ? sgpr_hard(1.0 / a.Mglobal) : sgpr(1.0 / a.Mglobal);
  float mean_f = 0.0f; double inv_denom = 1.0;
  if (ROLE == 0) {
    mean_f = sgpr((float)a.adv_ms[2 * a.mb]);
    const double idn = 1.0 / ((double)(float)a.adv_ms[2 * a.mb + 1] + 1e-8);
    inv_denom = X2 ? sgpr_hard(idn) : sgpr(idn);
  }
  const float eps = c.clip, lo = 1.0f - c.clip, hi = 1.0f + c.clip;
  const float u_exact = EXACT ? sgpr((float)a.vfix[0]) : 0.0f;
  const double nwin = EXACT ? sgpr(a.vfix[1]) : 0.0;
  const double entk0 = (double)c.ent_coeff / ((double)A * a.Mglobal), vk0 = (double)c.v_coef * 0.5 * invM;
  const double entk = X2 ? sgpr_hard(entk0) : sgpr(entk0);
  const double vk = X2 ? sgpr_hard(vk0) : sgpr(vk0);

  // A tile's records are loaded at the top of the tile: the minibatch is a contiguous slab (records.hip), so the loads are
  // L2 / HBM streaming reads whose latency the partner wave covers. (Loading one tile ahead kept seven more registers live
  // through the backward pass — spills — for no measurable gain once the gather through the permutation was gone.)
  int tile = rb * RW + wave, tstride = nwaves;
  if (a.xcd_align) {
    // Workgroups go to the 8 XCDs round-robin by index, and each XCD has its own L2. Actor and critic read the SAME 64-byte record of a
    // sample (different quarters), from different blocks: with the plain striding the two readers of a tile sit on different XCDs and
    // the record comes from HBM twice (FETCH_SIZE 252 MB per launch for 134 MB of records). Here tile t belongs to XCD t % 8 in both
    // roles — block counts are multiples of 8, so a role-local block index ≡ its launch index (mod 8) — and both roles walk their XCD's
    // tiles in the same order at the same relative pace: the second reader finds the line in L2.
    const int gx = rb & 7;
    tile = gx + 8 * ((rb >> 3) * RW + wave);
    tstride = 8 * (a.nblk[ROLE] >> 3) * RW;
  }
  Gathered<D> cur;
  constexpr bool PF = ((X2 && ABL == 0) || (X3 && ABL == 128 && CRL_PF_X3 != 0)) && !EXACT && CRL_PF_ENABLED != 0;   // (X3 with ABL 128 = update_x3_kernel: the gemm = 1 option kernel)
  const f32x4* pf4 = nullptr;
  const int* pfi = nullptr;
  unsigned pf_lds = 0;
  // position of this lane's sample in tile t (clamped: the lanes past the end of the last tile fetch sample 0 and are masked later)
  auto pos_of = [&](int t) { const int p = t * TILE + j; return p < M ? p : 0; };
  auto issue_dma = [&](int idx) {
    // the lane's half is recomputed here (two v_mbcnt) rather than kept: a loop-invariant per-lane base pointer was the value that
    // got spilled out of the 256-register budget
    unsigned ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const char* g = reinterpret_cast<const char*>(a.recs) + ((size_t)(unsigned)idx * 64u + (ln >= 32u ? (ROLE == 0 ? 16u : 32u) : 0u));
    unsigned keep;
    // lgkmcnt(0): the ds_reads of the slot's previous content have landed before the DMA may overwrite it; M0 = destination base,
    // written in the same statement that uses it (the compiler reserves M0) and restored
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(pf_lds) : "memory");
  };
  auto issue_perm_dma = [&](int t) {   // perm entries of tile t → the slot's last 256 bytes (both lane halves fetch the same entry)
    const int* g = a.perm + pos_of(t);
    unsigned keep;
    // (the destination offset goes into M0: an instruction offset would also be added to the GLOBAL address)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(pf_lds + 1024u) : "memory");
  };
  if constexpr (PF) {
    float* slot = pfslots + wave * PF_SLOT_FLOATS;
    pf4 = reinterpret_cast<const f32x4*>(slot);
    pfi = reinterpret_cast<const int*>(slot + 256);
    pf_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)slot);
    if (tile < ntiles) issue_dma(a.perm[pos_of(tile)]);
    if (tile + tstride < ntiles) issue_perm_dma(tile + tstride);
  }
  const float Gdw = X2 ? sgpr(a.dscale[ROLE]) : 1.0f;   // fp16x2 weight-gradient scale of this launch (mlp_x2.hpp)
  // fp16x2: G rides on the head cotangent (NOUT multiplies per tile), so δ2 exists only as δ2·G — exactly what the weight-gradient
  // operand wants (32 multiplies per tile saved); the backward-data product and db2 take the exact power of two back out
  const float invG = X2 ? sgpr(1.0f / Gdw) : 1.0f;
  float d2run = 0.0f;
  // (only where a wave has 16 tiles or more: with the 4-8 tiles per wave of a 4096 / 8192-env shard the same rule measured 4-5 % SLOWER)
  const bool balance = X2 && (ntiles >= 16 * tstride || a.prio_mode == 1);
  const int small_prio = (X2 && ntiles < 16 * tstride) ? a.prio_mode : 0;
  if (small_prio == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  int tile_parity = 0;
#ifdef CRL_EXP_STAMPS
  int tcount = 0;
#endif
  for (; tile < ntiles; tile += tstride) {
#ifdef CRL_EXP_STAMPS
    const bool ts_on = (tcount++ == 20);
    unsigned ts[6];
#endif
    CRL_TS(0);
    int partner_tile = 0;
    if (small_prio == 3) { if (((tile_parity++) & 1) == (wave >> 2)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    if (balance) {
      // The SIMD's arbiter favours the older of its two waves (w over w + 4): left alone, waves 0-3 finish their tiles a quarter of
      // the launch early (measured with timestamps: 395 of 531 µs) and waves 4-7 run the rest alone, at half the SIMD's issue rate.
      // Feedback instead of a fixed pattern (alternating the favour tile by tile gained half as much): every wave posts the tile it
      // starts (partners begin less than half a stride apart and advance by the same stride, so the index is the progress); whoever
      // is behind its SIMD partner asks for priority, whoever is ahead yields. Timing only — which wave works on which tile does
      // not change, so the gradients keep their bits. All waves now end within 1 % of each other: 0.536 → 0.512 ms per launch.
      // The partner's counter is read here and used after the tile's other LDS reads (one wait for all of them).
      volatile int* prog = reinterpret_cast<volatile int*>(scratch + RW * SCR);   // 8 counters where the staging flag was
      prog[wave] = tile;
      partner_tile = prog[wave ^ 4];
    }
    const int pos = tile * TILE + j;
    const bool ok = pos < M;
    // the Float64 role constants stay in scalar registers: re-pinned every tile, so that the compiler cannot hoist vector copies of
    // them out of the loop (such a copy of inv_denom was spilled, and its reload's vmcnt(0) drained the record prefetch mid-tile)
    double invM_t = invM, inv_denom_t = inv_denom, entk_t = entk, vk_t = vk;
    if constexpr (X2) asm volatile("" : "+s"(invM_t), "+s"(inv_denom_t), "+s"(entk_t), "+s"(vk_t));
    if constexpr (PF) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's records (issued a tile ago) and idx_next have arrived
      const f32x4 xv = pf4[j], q = pf4[32 + j];
      cur.x[0] = xv[0]; cur.x[1] = xv[1]; cur.x[2] = xv[2]; cur.x[3] = xv[3];
      if (ROLE == 0) { cur.act = __float_as_int(q[0]); cur.f0 = q[1]; cur.f1 = q[2]; }
      else { cur.act = 0; cur.f0 = q[0]; cur.f1 = q[1]; }
      if (tile + tstride < ntiles) issue_dma(pfi[lane]);     // wave-uniform branch; the entry came in with this tile's records
      if (tile + 2 * tstride < ntiles) issue_perm_dma(tile + 2 * tstride);
    } else {
      gather<D, ROLE>(a, ok ? pos : 0, cur);
    }
    if (balance) {
      const int d = __builtin_amdgcn_readfirstlane(tile - partner_tile), half = tstride >> 1;
      if (d < -half) __builtin_amdgcn_s_setprio(3); else if (d < half && (wave >> 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    CRL_TS(1);
    float x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = cur.x[i];
    f32x16 h1[2], h2[2];
    float out[NOUT], dout[NOUT];
    // opaque per-iteration offset: keeps the loop-invariant weight reads in LDS instead of hoisted into ~130 registers
    int lds_off = 0;
    asm volatile("" : "+v"(lds_off));
    const float* img = img0 + lds_off;
    // the same opaque offset on the wave's scratch tile: its per-lane row / column addresses are rebuilt every tile (a few VALU
    // instructions) instead of living in registers across the whole loop, where they were the values that got spilled
    float* T = T0 + lds_off;
    float* xs = T + 64 * TSTRIDE;
    float* d3s = xs + TILE * D;
    if constexpr (X2) mlp_forward_x2<D, NOUT>(img, x, h1, h2, out, lane);   // h1 = 2^14·tanh(…) from here on
    else if constexpr (X3) mlp_forward_x3<D, NOUT, true, ABL>(img, x, h1, h2, out, lane);   // ABL bit 128 (update_x3_kernel): the exp2 activation (mlp_x3.hpp)
    else mlp_forward<D, NOUT, true>(img, x, h1, h2, out, lane);
    CRL_TS(2);

    if constexpr ((ABL & 32) != 0) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i) dout[i] = out[i] * 1e-6f + cur.f0 + cur.f1 + (float)cur.act;
    } else if constexpr (ROLE == 0) {
      // policy loss + entropy (ppo.jl:213,219-228,242)
      float pr[A], lp[A];
      softmax_logsoftmax<A>(out, pr, lp);
      const int act = cur.act;
      float nlp = lp[0];
#pragma unroll
      for (int i = 1; i < A; ++i) nlp = (act == i) ? lp[i] : nlp;
      double Hs = 0.0;
#pragma unroll
      for (int i = 0; i < A; ++i) Hs += (double)(-(pr[i] * lp[i]));
      const double Ahat = (double)(cur.f1 - mean_f) * inv_denom_t;
      const float ratio = expf(nlp - cur.f0);
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const double pg1 = -Ahat * (double)ratio, pg2 = -Ahat * (double)rc;
      double dnlp, pg;
      if (pg1 > pg2) { pg = pg1; dnlp = pg1; }
      else { pg = pg2; dnlp = (ratio >= lo && ratio <= hi) ? pg1 : 0.0; }
      dnlp *= invM_t;
#pragma unroll
      for (int i = 0; i < A; ++i)
        dout[i] = (float)(dnlp * ((i == act ? 1.0 : 0.0) - (double)pr[i]) + entk_t * (double)pr[i] * ((double)lp[i] + Hs));
      if (ok && hf == 0) { ls0 += pg; ls1 += Hs; }
    } else {
      // value loss (ppo.jl:214,231-240)
      const float v = out[0], R = cur.f1, ov = cur.f0;
      double dv, term;
      if (c.clip_vloss) {
        const float dvv = v - ov;
        const float cl = fminf(fmaxf(dvv, -eps), eps);
        const float vc = ov + cl;
        const float q = (vc - R) * (vc - R);
        const bool q_wins = EXACT ? !(u_exact > q) : true;  // max.(u, q): ties → q
        term = q_wins ? (double)q : (double)u_exact;
        const double inner = (q_wins && dvv >= -eps && dvv <= eps) ? 2.0 * (double)(vc - R) : 0.0;
        dv = vk_t * (nwin * invM_t + inner);
      } else {
        const float e = v - R;
        term = (double)(e * e);
        dv = vk_t * 2.0 * (double)e;
      }
      dout[0] = (float)dv;
      if (ok && hf == 0) {
        ls0 += (double)(v - R * R);
        ls1 += term;
        if (!EXACT) a.newv[pos] = v;
      }
    }
    if (!ok) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i) dout[i] = 0.0f;
    }

    CRL_TS(3);
    // ---- backward ------------------------------------------------------------------------------------
    // (1) h2ᵀ, the output cotangents and x into the wave-private scratch
    store_transposed(T, h2, j, hf);
    if (hf == 0) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i) d3s[i * TILE + j] = dout[i];
#pragma unroll
      for (int i = 0; i < D; ++i) xs[j * D + i] = x[i];
    }
    CRL_PHASE();
    if constexpr (!(ABL & 8))
    // (2) lane = row: dW3[a][lane] += Σ_s h2[lane][s]·δ3[a][s]. The LDS reads of half a row are issued together and waited
    // for with counted lgkmcnt (a read → wait → use chain per quad exposed the full LDS latency eight times); two halves
    // keep the batch at 12 registers-quads so nothing spills. db3 is a per-lane sum, folded over lanes once per kernel.
    {
      const f32x4* tr = reinterpret_cast<const f32x4*>(T + lane * TSTRIDE);
      float accw[NOUT], ow3[NOUT], ob3[NOUT];
#pragma unroll
      for (int i = 0; i < NOUT; ++i) { accw[i] = 0.0f; ow3[i] = acc_begin(K_W3 + i); ob3[i] = acc_begin(K_B3 + i); }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 rq[4], dv[NOUT][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rq[q] = tr[4 * half + q];
#pragma unroll
        for (int i = 0; i < NOUT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) dv[i][q] = reinterpret_cast<const f32x4*>(d3s + i * TILE)[4 * half + q];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < NOUT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) accw[i] = __builtin_fmaf(rq[q][e], dv[i][q][e], accw[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < NOUT; ++i) { acc_end(K_W3 + i, ow3[i] + accw[i]); acc_end(K_B3 + i, ob3[i] + (hf == 0 ? dout[i] : 0.0f)); }
    }
    CRL_TS(4);
    // (3) δ2 = (W3ᵀ·δ3) ⊙ (1 − h2²) in C-fragment registers (h2 dies here)
    f32x16 d2[2];
    {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[mt][r] = 0.0f;
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const f32x4* w = reinterpret_cast<const f32x4*>(img + I::W3 + i * 64 + hf * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 wv = w[q];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int idx = q * 4 + e;
            d2[idx >> 4][idx & 15] = __builtin_fmaf(wv[e], X2 ? dout[i] * Gdw : dout[i], d2[idx >> 4][idx & 15]);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[mt][r] *= (1.0f - h2[mt][r] * h2[mt][r]);
    }
    CRL_TS(5);
    // (4) dh1 = W2ᵀ·δ2 (A-fragments of W2ᵀ from LDS, B = δ2 registers); δ1 = dh1 ⊙ (1 − h1²)
    f32x16 d1[2];
    float d2max = 0.0f;   // fp16x2: this sample's largest |δ2|
    float d1f = 1.0f;     // fp16x2: this sample's backward-data unscale
    {
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c0[r] = 0.0f; c1[r] = 0.0f; }
      if constexpr ((ABL & 16) != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { c0[r] = d2[0][r]; c1[r] = d2[1][r]; }
      } else if constexpr (X2) {
        // each sample's cotangent column scaled by its own power of two into the fp16 window, unscaled below
        float sc, sinv;
        sample_scale(d2, sc, sinv, d2max);
        d2run = __builtin_fmaxf(d2run, d2max);
        f32x16 ds[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) ds[mt][r] = d2[mt][r] * sc;
        dense64_x2(img + I::WB2H, ds, c0, c1, lane);
        d1f = sinv * ((1.0f / X2_W_SCALE) * invG);   // the unscale (per-sample scale, 2^8 of the weights, G) rides in the (1 − h1²) factor below
      } else if constexpr (X3) {
        dense64_x3(img + I::WB2P, d2, c0, c1, lane);
      } else {
        const f32x4* w0 = reinterpret_cast<const f32x4*>(img + I::WB2) + lane;
        const f32x4* w1 = reinterpret_cast<const f32x4*>(img + I::WB2 + 2048) + lane;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const f32x4 fa = w0[s4 * 64], fb = w1[s4 * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int s = s4 * 4 + e;
            const float b = d2[s >> 4][s & 15];
            c0 = mfma32(fa[e], b, c0);
            c1 = mfma32(fb[e], b, c1);
          }
        }
      }
      if constexpr (X2 && !(ABL & 16)) {
        // δ1 = (c·f)·(1 − h1²) with h1 carried as 2^14·h1 and f the per-sample unscale: c·(f − (h1s·kf)·h1s), three instructions
        const float kf = d1f * (1.0f / (X2_ACT_SCALE * X2_ACT_SCALE));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          d1[0][r] = c0[r] * __builtin_fmaf(-(h1[0][r] * kf), h1[0][r], d1f);
          d1[1][r] = c1[r] * __builtin_fmaf(-(h1[1][r] * kf), h1[1][r], d1f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          constexpr float k = X2 ? 1.0f / (X2_ACT_SCALE * X2_ACT_SCALE) : 1.0f;   // h1 is carried as 2^14·h1 in the fp16x2 flavour
          d1[0][r] = c0[r] * (1.0f - (h1[0][r] * k) * h1[0][r]);
          d1[1][r] = c1[r] * (1.0f - (h1[1][r] * k) * h1[1][r]);
        }
      }
    }
    CRL_PHASE();
    CRL_TS(6);
    // (7) runs before (5)/(6): δ1 dies here, so the weight-gradient phase below holds 32 fewer live registers (no spills)
    if constexpr (!(ABL & 8)) {
    // (7) δ1ᵀ → scratch; lane = row: db1, dW1[lane][c] += Σ_s δ1[lane][s]·x[s][c]
    store_transposed(T, d1, j, hf);
    CRL_PHASE();
    {
      const f32x4* tr = reinterpret_cast<const f32x4*>(T + lane * TSTRIDE);
      float sb = acc_begin(K_B1), w1[4] = {acc_begin(0), acc_begin(1), acc_begin(2), acc_begin(3)};
      f32x4 t4[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t4[q] = tr[q];
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) {          // the tile's observations, 8 samples (8 broadcast reads) at a time
        f32x4 xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = *reinterpret_cast<const f32x4*>(xs + (8 * q2 + e) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = t4[2 * q2 + (e >> 2)][e & 3];
          sb += dv;
#pragma unroll
          for (int i = 0; i < 4; ++i) w1[i] = __builtin_fmaf(dv, xv[e][i], w1[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      acc_end(K_B1, sb);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc_end(i, w1[i]);
    }
    CRL_PHASE();
    } else { acc_end(K_B1, acc_begin(K_B1) + d1[0][0] + d1[1][5]); }
    CRL_TS(7);
    if constexpr (!(ABL & 2)) {
    // (5) δ2ᵀ → scratch; db2; B-fragments (δ2 rows on lanes, samples along k: smp(s,hf) = s + 16hf) (δ2 dies here)
    store_transposed(T, d2, j, hf);
    CRL_PHASE();
    f32x4 bfr[2][4];
    f32x4 braw[2][2][2];  // x3: raw δ2ᵀ B-fragments [ni][ks][half] (split into bf16 pieces at use: 32 registers, not 48)
    {
      const f32x4* tr = reinterpret_cast<const f32x4*>(T + lane * TSTRIDE);
      float s = acc_begin(K_B2);
#pragma unroll
      for (int q = 0; q < 8; ++q) { const f32x4 t4 = tr[q]; s += (t4[0] + t4[1]) + (t4[2] + t4[3]); }
      acc_end(K_B2, s);
      if constexpr (X3) {
        // k-step ks covers samples 16ks + 8hf + (0..7): two b128 reads per fragment, split into bf16 pieces
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const f32x4* fr = reinterpret_cast<const f32x4*>(T + (32 * ni + j) * TSTRIDE + 16 * ks + 8 * hf);
            braw[ni][ks][0] = fr[0]; braw[ni][ks][1] = fr[1];
          }
      } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f32x4* fr = reinterpret_cast<const f32x4*>(T + (32 * ni + j) * TSTRIDE + 16 * hf);
#pragma unroll
          for (int q = 0; q < 4; ++q) bfr[ni][q] = fr[q];
        }
      }
    }
    CRL_PHASE();
    CRL_TS(8);
    // (6) h1ᵀ → scratch (h1 dies here); A-fragments streamed; dW2ᵀ[mj][ni] += h1[mj-block]·δ2[ni-block]ᵀ over 32 samples
    store_transposed(T, h1, j, hf);
    CRL_PHASE();
#ifdef CRL_COUNT_PROBE
    // scripts/count_isa.py builds this file with the cold paths (in-loop bf16x3 weight gradient, per-role bf16x3 fallback) compiled
    // out, so that each role's tile loop is one plain loop whose instructions can be counted; never part of the library
    if (X2) {
#else
    if (X2 && dw_tile_fits(d2max, 1.0f)) {   // d2max is the largest |δ2·G| of the tile
#endif
      if constexpr (X2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          P2 ap[2], bp[2];
#pragma unroll
          for (int mj = 0; mj < 2; ++mj) {
            const f32x4* fr = reinterpret_cast<const f32x4*>(T + (32 * mj + j) * TSTRIDE + 16 * ks + 8 * hf);
            const f32x4 f0 = fr[0], f1 = fr[1];
            const float xa[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
            ap[mj] = split2(xa);
          }
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f32x4 f0 = braw[ni][ks][0], f1 = braw[ni][ks][1];
            const float xb[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};   // δ2·G already
            bp[ni] = split2(xb);
          }
          dW2t[0][0] = mfma_x2(ap[0], bp[0], dW2t[0][0]);
          dW2t[0][1] = mfma_x2(ap[0], bp[1], dW2t[0][1]);
          dW2t[1][0] = mfma_x2(ap[1], bp[0], dW2t[1][0]);
          dW2t[1][1] = mfma_x2(ap[1], bp[1], dW2t[1][1]);
        }
      }
    } else if constexpr (X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        P3 ap[2];
#pragma unroll
        for (int mj = 0; mj < 2; ++mj) {
          const f32x4* fr = reinterpret_cast<const f32x4*>(T + (32 * mj + j) * TSTRIDE + 16 * ks + 8 * hf);
          const f32x4 f0 = fr[0], f1 = fr[1];
          const float xa[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
          ap[mj] = split3(xa);
        }
        P3 bp[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f32x4 f0 = braw[ni][ks][0], f1 = braw[ni][ks][1];
          const float xb[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};   // δ2·G already (G = 1 outside the fp16x2 flavour)
          bp[ni] = split3(xb);
        }
        dW2t[0][0] = mfma_x3(ap[0], bp[0], dW2t[0][0]);
        dW2t[0][1] = mfma_x3(ap[0], bp[1], dW2t[0][1]);
        dW2t[1][0] = mfma_x3(ap[1], bp[0], dW2t[1][0]);
        dW2t[1][1] = mfma_x3(ap[1], bp[1], dW2t[1][1]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(T + (j) * TSTRIDE + 16 * hf + 4 * q);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(T + (32 + j) * TSTRIDE + 16 * hf + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dW2t[0][0] = mfma32(a0[e], bfr[0][q][e], dW2t[0][0]);
          dW2t[0][1] = mfma32(a0[e], bfr[1][q][e], dW2t[0][1]);
          dW2t[1][0] = mfma32(a1[e], bfr[0][q][e], dW2t[1][0]);
          dW2t[1][1] = mfma32(a1[e], bfr[1][q][e], dW2t[1][1]);
        }
      }
    }
    CRL_PHASE();
    }
#ifdef CRL_EXP_STAMPS
    CRL_TS(9);
    if (ts_on && (threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 10; ++k) if (crl_ts_slot(k) >= 0) crl_dbg_tstamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + k] = ts[crl_ts_slot(k)];
      crl_dbg_tstamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + 15] = (unsigned)ROLE + 1u;
    }
#endif
  }

  CRL_STAMP(2);
  // fp16x2: the weight-gradient accumulators carry 2^14 (h1) · G (δ2); the launch's largest |δ2| goes to the next launch's G
  const float dw_unscale = X2 ? (1.0f / X2_ACT_SCALE) / Gdw : 1.0f;
  if constexpr (X2) {
    float m = d2run;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0 && m > 0.0f) atomicMax(a.dmax + ROLE, __float_as_uint(m * invG));   // d2run tracked |δ2·G|
  }
  // ---- block reduction: waves add their accumulators into one LDS image in flat Flux order ------------------
  if constexpr (LACC) {
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NACC; ++i) racc[i] = ACC[64 * i];
  }
  if constexpr (X2) racc[K_B2] *= invG;   // db2 summed δ2·G
  __syncthreads();
  // one wave's accumulators into an LDS image in flat Flux order (ADD = false: plain stores — a wave writes every entry)
  auto deposit = [&](float* R, auto add) {
    constexpr bool ADD = decltype(add)::value;
#pragma unroll
    for (int mj = 0; mj < 2; ++mj)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = R + P::W2 + (32 * ni + j) + H * (32 * mj + rowmap(r, hf));
          const float v = dW2t[mj][ni][r] * dw_unscale;
          *q = ADD ? *q + v : v;
        }
#pragma unroll
    for (int i = 0; i < D; ++i) { float* q = R + P::W1 + lane + H * i; *q = ADD ? *q + racc[i] : racc[i]; }
    { float* q = R + P::B1 + lane; *q = ADD ? *q + racc[K_B1] : racc[K_B1]; }
    { float* q = R + P::B2 + lane; *q = ADD ? *q + racc[K_B2] : racc[K_B2]; }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) { float* q = R + P::W3 + i + NOUT * lane; *q = ADD ? *q + racc[K_W3 + i] : racc[K_W3 + i]; }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const float b3 = wave_sum(racc[K_B3 + i]);
      if (lane == 0) { float* q = R + P::B3 + i; *q = ADD ? *q + b3 : b3; }
    }
  };
  float* gp = a.gpart + ((size_t)ROLE * a.pmax + rb) * a.gstride;
  double* lsum;
  if constexpr (X2 && RW == 8) {
    // Four images in the dead weight image + scratch tiles: waves 0-3 STORE theirs, waves 4-7 add on top, then one pass folds the
    // four into the block's partial — two serial rounds instead of eight (the eight-round version was ≈14 µs of every launch:
    // 2.5 % at the headline size, 15 % of the 90 µs launch of an 8192-env shard). Fixed order: ((w0+w4) + (w1+w5)) + ((w2+w6) + (w3+w7)).
    constexpr int IMG = ((P::SIZE + 3) / 4) * 4;
    static_assert(4 * IMG + 64 <= NetImageX2<D, NOUT>::SIZE + 8 * SCR_FLOATS_X3, "reduction images do not fit the dead LDS");
    lsum = reinterpret_cast<double*>(smem + 4 * IMG);
    ls0 = wave_sum(ls0); ls1 = wave_sum(ls1);
    if (lane == 0) { lsum[wave] = ls0; lsum[8 + wave] = ls1; }
    if (wave < 4) deposit(smem + wave * IMG, std::false_type{});
    __syncthreads();
    if (wave >= 4) deposit(smem + (wave - 4) * IMG, std::true_type{});
    __syncthreads();
    for (int i = tid; i < P::SIZE; i += NT) gp[i] = (smem[i] + smem[IMG + i]) + (smem[2 * IMG + i] + smem[3 * IMG + i]);
  } else {
    float* R = smem;  // the weight image is dead now
    for (int i = tid; i < P::SIZE; i += NT) R[i] = 0.0f;
    lsum = reinterpret_cast<double*>(smem + 6144);  // inside the dead weight image, past R (all LDS stays dynamic)
    ls0 = wave_sum(ls0); ls1 = wave_sum(ls1);
    if (lane == 0) { lsum[wave] = ls0; lsum[8 + wave] = ls1; }
    __syncthreads();
    for (int w = 0; w < RW; ++w) {
      if (wave == w) {
#pragma unroll
        for (int mj = 0; mj < 2; ++mj)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              R[P::W2 + (32 * ni + j) + H * (32 * mj + rowmap(r, hf))] += dW2t[mj][ni][r] * dw_unscale;
#pragma unroll
        for (int i = 0; i < D; ++i) R[P::W1 + lane + H * i] += racc[i];
        R[P::B1 + lane] += racc[K_B1];
        R[P::B2 + lane] += racc[K_B2];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) R[P::W3 + i + NOUT * lane] += racc[K_W3 + i];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
          const float b3 = wave_sum(racc[K_B3 + i]);
          if (lane == 0) R[P::B3 + i] += b3;
        }
      }
      __syncthreads();
    }
    for (int i = tid; i < P::SIZE; i += NT) gp[i] = R[i];
  }
  CRL_STAMP(3);
  if (tid == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int w = 0; w < RW; ++w) { s0 += lsum[w]; s1 += lsum[8 + w]; }
    double* lp = a.lpart + ((size_t)ROLE * a.pmax + rb) * 2;
    lp[0] = s0; lp[1] = s1;
  }
  return true;
}

// Speculative pass (assumes u <= 0, see header). Waves w and w+4 of a block share a SIMD.
// bf16x3 flavour (mlp_x3.hpp): the split weight images are 51 KB per network, so a 512-thread block carries ONE role
// (blocks [0, nblk[0]) = actor, the rest = critic) and its 8 waves share that image.
// LDS of the main-pass kernels (both flavours are launched with it): the bf16x3 layout — the larger one, which the fp16x2 kernel falls back to — then 8 record-prefetch slots
constexpr int X2_KERNEL_LDS_FLOATS = NetImageX3<4, 2, true>::SIZE + 8 * SCR_FLOATS_X3 + 12;   // + the staging flag / the 8 progress counters   // the bf16x3 fallback's layout is the larger one
static_assert(X2_KERNEL_LDS_FLOATS % 4 == 0, "prefetch slots are 16-byte aligned");
static_assert(PF_SLOT_FLOATS % 4 == 0 && (X2_KERNEL_LDS_FLOATS + 8 * PF_SLOT_FLOATS) * 4 <= 160 * 1024, "update_x2_kernel's LDS exceeds a CU's 160 KB");
template <int D, int A>
__global__ void __launch_bounds__(512, 2) update_x3_kernel(UpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // Option gemm = 1 (bench.py `strict_f32`: 24-bit operands by construction) with the update pass's exp2 activation, like the fp16x2 default — the option is about the
  // operand width of the products, not about which approximation of tanh runs: 0.81 -> 0.73 ms per launch at M = 2,097,152, parity margins unchanged
  // (profiles/r06_parity_margins.json). The bf16x3 FALLBACKS below (a weight outside the fp16 window, a weight-gradient scale miss) keep NNlib's rational tanh_fast: they
  // run exactly when weights are extreme and units saturate, where 1 - h² of the two approximations differs in relative terms
  // (test_options_are_validated_and_fallback_is_automatic: a weight of 300).
  float* pfslots = smem + X2_KERNEL_LDS_FLOATS;   // the record prefetch of the fp16x2 kernel (-1 % there): wired here as well, off (CRL_PF_X3)
  if ((int)blockIdx.x < a.nblk[0]) update_role<D, A, 0, false, true, 8, 128>(a, blockIdx.x, smem, smem + NetImageX3<D, A, true>::SIZE, pfslots);
  else update_role<D, A, 1, false, true, 8, 128>(a, blockIdx.x - a.nblk[0], smem, smem + NetImageX3<D, 1, true>::SIZE, pfslots);
}
// After update_t16_kernel (update16.hpp): nothing unless a tile of that launch did not fit the carried weight-gradient scale (or a weight left the fp16
// window) — then the whole minibatch is recomputed on bf16x3 (no range limits) into the same partial buffers, before the reduce reads them.
template <int D, int A>
__global__ void __launch_bounds__(512, 2) update_repair_kernel(UpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (*a.dw_miss == 0u) return;
  if ((int)blockIdx.x < a.nblk[0]) update_role<D, A, 0, false, true, 8>(a, blockIdx.x, smem, smem + NetImageX3<D, A, true>::SIZE);
  else update_role<D, A, 1, false, true, 8>(a, blockIdx.x - a.nblk[0], smem, smem + NetImageX3<D, 1, true>::SIZE);
}
}  // namespace crl
#include "update16.hpp"   // the 16-sample-tile flavour (needs sgpr / CRL_PHASE above)
namespace crl {
// fp16x2 flavour (the default): forward / backward-data products on the f16 matrix pipe, three MFMAs per product
template <int D, int A>
__global__ void __launch_bounds__(512, 2) update_x2_kernel(UpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // a role whose hidden-layer weights left the fp16 window (|w| >= 255) runs as bf16x3 for this launch: same operands, no range limit
  float* pfslots = smem + X2_KERNEL_LDS_FLOATS;   // 8 x 1 KB record-prefetch slots behind the larger (bf16x3) layout
#ifdef CRL_COUNT_PROBE
  if ((int)blockIdx.x < a.nblk[0]) update_role<D, A, 0, false, true, 8, 0, true>(a, blockIdx.x, smem, smem + NetImageX2<D, A>::SIZE, pfslots);
  else update_role<D, A, 1, false, true, 8, 0, true>(a, blockIdx.x - a.nblk[0], smem, smem + NetImageX2<D, 1>::SIZE, pfslots);
  return;
#endif
  if ((int)blockIdx.x < a.nblk[0]) {
    if (!update_role<D, A, 0, false, true, 8, 0, true>(a, blockIdx.x, smem, smem + NetImageX2<D, A>::SIZE, pfslots))
      update_role<D, A, 0, false, true, 8>(a, blockIdx.x, smem, smem + NetImageX3<D, A, true>::SIZE);
  } else {
    if (!update_role<D, A, 1, false, true, 8, 0, true>(a, blockIdx.x - a.nblk[0], smem, smem + NetImageX2<D, 1>::SIZE, pfslots))
      update_role<D, A, 1, false, true, 8>(a, blockIdx.x - a.nblk[0], smem, smem + NetImageX3<D, 1, true>::SIZE);
  }
}
// Exact critic-only pass with the known scalar u and count; runs only when stats_kernel raised the flag
template <int D, int A>
__global__ void __launch_bounds__(256, 2) update_vfix_kernel(UpdateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.vfix[3] == 0.0) return;  // speculation held: nothing to redo
  update_role<D, A, 1, true, false, 4>(a, blockIdx.x, smem, smem + NetImage<D, 1, true>::SIZE);
}

// Σ over per-block partials in fixed order → flat gradient (+ the loss sums appended for the all-reduce message)
// msg layout: [P gradient floats][pg_sum, ent_sum, u_sum, q_sum as floats]
// Block = 64 consecutive outputs x RG groups of partials (group g sums blocks b ≡ g mod RG in order, then the groups are
// folded in order): 256-B coalesced rows and every partial of an output in flight at once — at 256 update blocks a
// thread issues its 8 loads back to back instead of walking 32 of them in 8 dependent rounds — still a fixed summation order.
constexpr int RG = 16;
template <int MODE>
__global__ void __launch_bounds__(64 * RG) reduce_kernel(const float* __restrict__ gpart, const double* __restrict__ lpart,
                                                         int nblkA, int nblkC, int pmax, int gstride, int Pa, int Pc,
                                                         float* __restrict__ msg, StatsArgs st) {
  const double* vfix = st.vfix;
  if (MODE == 1 && vfix[3] == 0.0) return;
  __shared__ float sm[RG][64];
  __shared__ double smd[RG][4];
  const int P = Pa + Pc;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.0f;
  bool live = i < P;
  int role = 0;
  if (live) {
    role = i >= Pa;
    if (MODE == 1 && role == 0) live = false;
  }
  if (live) {
    const float* gp = gpart + (size_t)role * pmax * gstride + (role ? i - Pa : i);
    const int blocks_per_role = role ? nblkC : nblkA;
    int b = g;
    for (; b + 3 * RG < blocks_per_role; b += 4 * RG) {
      const float v0 = gp[(size_t)b * gstride], v1 = gp[(size_t)(b + RG) * gstride], v2 = gp[(size_t)(b + 2 * RG) * gstride],
                  v3 = gp[(size_t)(b + 3 * RG) * gstride];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < blocks_per_role; b += RG) s += gp[(size_t)b * gstride];
  }
  sm[g][o] = s;
  // the four loss sums ride on the last block: thread (which = o < 4, group g)
  double ds = 0.0;
  const bool last = blockIdx.x == gridDim.x - 1;
  if (last && o < 4) {
    const int which = o, lrole = which >> 1;  // 0 pg, 1 ent (actor) ; 2 u, 3 q (critic)
    if (!(MODE == 1 && lrole == 0)) {
      const double* l = lpart + (size_t)lrole * pmax * 2 + (which & 1);
      const int nb = lrole ? nblkC : nblkA;
      for (int b = g; b < nb; b += RG) ds += l[b * 2];
    }
    smd[g][o] = ds;
  }
  __syncthreads();
  if (g == 0 && live) {
    float t = sm[0][o];
#pragma unroll
    for (int q = 1; q < RG; ++q) t += sm[q][o];
    msg[i] = t;
  }
  if (last && g == 0 && o < 4 && !(MODE == 1 && (o >> 1) == 0)) {
    double t = smd[0][o];
#pragma unroll
    for (int q = 1; q < RG; ++q) t += smd[q][o];
    msg[P + o] = (float)t;
  }
  if (MODE == 0 && last && st.dscale && threadIdx.x < 2) {
    // fp16x2 weight gradient: this launch's largest |δ2| becomes the next launch's scale (mlp_x2.hpp), per role
    unsigned* mx = reinterpret_cast<unsigned*>(st.dscale + 2);
    st.dscale[threadIdx.x] = dw_next_scale(mx[threadIdx.x], st.dscale[threadIdx.x]);
    mx[threadIdx.x] = 0u;
    if (threadIdx.x == 0) mx[2] = 0u;   // the 16-sample kernel's miss flag: update_repair_kernel has run by now
  }
  if (last && st.fused) {
    __syncthreads();  // the four sums written above are visible to thread 0 of this block
    if (threadIdx.x == 0) compute_stats(msg, P, st.c, st.Mglobal, st.adv_ms, st.mb, st.vfix, st.out, MODE);
  }
}

// reduce_kernel<0> + Optimiser(ClipNorm(0.5), Adam(η)) in ONE launch (single GPU, speculative step: nothing sits between the
// gradient and the optimiser). Same blocks, same fixed summation order as reduce_kernel; then every block leaves Σg² of its 64
// outputs per parameter array, the grid meets at a ticket (144 blocks of 1024 threads: all resident — nothing else runs on the
// device between an update kernel and the next one), every block sums the partials of the arrays it touches in block order
// (all blocks that touch an array compute the same norm bit for bit) and applies ClipNorm + Adam to its 64 entries with the
// arithmetic of clipnorm_adam_kernel (optim.hip; oracle: orc_clipnorm_adam). The β powers are read before the ticket and
// advanced after it by the block that holds an array's first entry. Saves a launch, a launch gap and the one-block-per-array
// walk of the 4,096-entry arrays per optimiser step.
struct FusedOptimArgs {
  int off[13];
  float* params; float* m; float* v; double* betap; double* part /* [12][gridDim.x] */; unsigned* ticket /* [0] arrivals, [1] sticky time-out flag */; unsigned target;
  unsigned long long timeout;   // ticks of the 100 MHz wall clock a block waits at the meeting point before it gives up (2 s)
  double eta, thresh;
};
// PEER = true (round 5): the DATA-PARALLEL optimiser step as one launch over the peer mailboxes of peer.hip — reduce → push this block's 64 sums into
// every rank's mailbox → flag → wait for the W flags of this chunk → add the W slots in rank order → Σg² → ticket → ClipNorm + Adam. Replaces
// reduce_kernel + peer_allreduce_kernel + clipnorm_adam_kernel (three launches, ≈ 40 µs beside an 85 µs update launch at 8192 envs per rank). A chunk
// is a block's 64 floats, so chunks never wait on each other across ranks; the grid-wide ticket stays local to the GPU. The four loss sums ride in the
// last block's chunk (indices P … P + 3), the statistics record is written from the all-reduced sums by that block. Every rank adds the same slots in
// the same order and clips by the same norms: replicas stay bit-identical.
template <bool PEER>
__global__ void __launch_bounds__(64 * RG) reduce_optim_kernel(const float* __restrict__ gpart, const double* __restrict__ lpart,
                                                               int nblkA, int nblkC, int pmax, int gstride, int Pa, int Pc,
                                                               float* __restrict__ msg, StatsArgs st, FusedOptimArgs oa, PeerArgs pa) {
#pragma clang fp contract(off)
  __shared__ float sm[RG][64];
  __shared__ double smd[RG][4];
  __shared__ double nrm2[12];
  __shared__ float gsum[4];
  const int P = Pa + Pc;
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.0f;
  const bool live = i < P;
  if (live) {
    const int role = i >= Pa;
    const float* gp = gpart + (size_t)role * pmax * gstride + (role ? i - Pa : i);
    const int blocks_per_role = role ? nblkC : nblkA;
    int b = g;
    for (; b + 3 * RG < blocks_per_role; b += 4 * RG) {
      const float v0 = gp[(size_t)b * gstride], v1 = gp[(size_t)(b + RG) * gstride], v2 = gp[(size_t)(b + 2 * RG) * gstride],
                  v3 = gp[(size_t)(b + 3 * RG) * gstride];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < blocks_per_role; b += RG) s += gp[(size_t)b * gstride];
  }
  sm[g][o] = s;
  double ds = 0.0;
  // the block whose chunk holds the loss sums (indices P … P + 3; the launcher sizes the grid so that one block holds all four)
  const bool last = blockIdx.x == (unsigned)(P / 64);
  if (last && o < 4) {
    const int which = o, lrole = which >> 1;
    const double* l = lpart + (size_t)lrole * pmax * 2 + (which & 1);
    const int nb = lrole ? nblkC : nblkA;
    for (int b = g; b < nb; b += RG) ds += l[b * 2];
    smd[g][o] = ds;
  }
  __syncthreads();
  // the arrays this block's 64 outputs belong to: [a_lo, a_hi]
  const int i_lo = blockIdx.x * 64, i_hi = (i_lo + 63 < P - 1) ? i_lo + 63 : P - 1;
  int a_lo = 0, a_hi = 0;
  while (a_lo < 11 && i_lo >= oa.off[a_lo + 1]) ++a_lo;
  while (a_hi < 11 && i_hi >= oa.off[a_hi + 1]) ++a_hi;
  int arr = a_lo;
  float grad = 0.0f;
  double bp0 = 0.0, bp1 = 0.0;
  float m_old = 0.0f, v_old = 0.0f, p_old = 0.0f;   // this entry's optimiser state: fetched before the meeting point, under its latency
  if (last && g == 0 && o < 4) {
    double t = smd[0][o];
#pragma unroll
    for (int q = 1; q < RG; ++q) t += smd[q][o];
    gsum[o] = (float)t;
  }
  if (last) __syncthreads();
  unsigned timed_out = 0;
  if (g == 0) {
    if (live) {
      m_old = oa.m[i]; v_old = oa.v[i]; p_old = oa.params[i];
      float t = sm[0][o];
#pragma unroll
      for (int q = 1; q < RG; ++q) t += sm[q][o];
      grad = t;
      while (arr < 11 && i >= oa.off[arr + 1]) ++arr;
      bp0 = oa.betap[2 * arr]; bp1 = oa.betap[2 * arr + 1];
    }
    const bool lsum = last && i >= P && i < P + 4;          // this lane carries one of the four loss sums
    float val = live ? grad : (lsum ? gsum[i - P] : 0.0f);
    if constexpr (PEER) {
      // ---- the exchange, one wave, one chunk (peer.hip's protocol at 64 floats per chunk)
      const bool carry = live || lsum;
      const int par = (int)(pa.seq & 1u), W = pa.world;
      for (int d = 1; d <= W; ++d) {                        // my chunk into my slot of every mailbox, right-hand neighbour first
        const int p = (pa.rank + d) % W;
        float* dst = reinterpret_cast<float*>(pa.box[p] + pa.data_off + ((size_t)par * W + pa.rank) * pa.slot_bytes);
        if (carry) __builtin_nontemporal_store(val, dst + i);
      }
      __threadfence_system();                               // every lane's stores are out before a flag of this chunk moves
      if (o < W) {
        uint32_t* f = reinterpret_cast<uint32_t*>(pa.box[o]) + ((size_t)par * W + pa.rank) * pa.nblk + blockIdx.x;
        __hip_atomic_store(f, pa.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t* gq = reinterpret_cast<const uint32_t*>(pa.box[pa.rank]) + ((size_t)par * W + o) * pa.nblk + blockIdx.x;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(gq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != pa.seq) {
          __builtin_amdgcn_s_sleep(2);
          if (wall_clock64() - t0 > pa.timeout_ticks) { atomicExch(pa.err, 1u); timed_out = 1; break; }
        }
      }
      timed_out = (unsigned)(__builtin_amdgcn_ballot_w64(timed_out != 0) != 0);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      if (carry) {
        // slots written by other devices while this kernel runs: system-scope atomic loads, served at the memory side (peer.hip)
        const char* mine = pa.box[pa.rank] + pa.data_off + (size_t)par * W * pa.slot_bytes;
        float t = 0.0f;
        for (int r = 0; r < W; ++r) {
          const unsigned bits = __hip_atomic_load(reinterpret_cast<const unsigned*>(mine + (size_t)r * pa.slot_bytes) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          t += __uint_as_float(bits);
        }
        val = t;
      }
      if (live) grad = val;
    }
    if (live || lsum) msg[i] = val;                         // comm_buf keeps the (all-reduced) message: CRL_F_GRADS, the statistics of a replay
    if (lsum) gsum[i - P] = val;
    const double gsq = live ? (double)grad * (double)grad : 0.0;
    for (int a = a_lo; a <= a_hi; ++a) {
      const double t = wave_sum((live && arr == a) ? gsq : 0.0);
      if (o == 0) __hip_atomic_store(oa.part + (size_t)a * gridDim.x + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (last && st.dscale && threadIdx.x < 2) {
    unsigned* mx = reinterpret_cast<unsigned*>(st.dscale + 2);
    st.dscale[threadIdx.x] = dw_next_scale(mx[threadIdx.x], st.dscale[threadIdx.x]);
    mx[threadIdx.x] = 0u;
    if (threadIdx.x == 0) mx[2] = 0u;   // the 16-sample kernel's miss flag: update_repair_kernel has run by now
  }
  if (last) {
    __syncthreads();
    if (threadIdx.x == 0) compute_stats4(gsum[0], gsum[1], gsum[2], gsum[3], st.c, st.Mglobal, st.adv_ms, st.mb, st.vfix, st.out, 0);
  }
  if (g != 0) return;                       // the optimiser half runs on the block's first wave
  // grid-wide meeting point. Everything that crosses blocks (the Σg² partials, the ticket) moves through agent-scope atomic
  // stores / loads, which are served at the device's coherent level: no release / acquire fences — on this GPU those write
  // back and invalidate a whole L2, which cost as much as the kernel boundary this launch exists to save (17.8 µs with fences).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "reduce_optim_kernel's fence-free meeting point relies on gfx942 / gfx950 behaviour (sc1 atomics are write-through and counted in vmcnt): port the ordering (release on the ticket add, acquire after the wait) before building for another target"
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partials have arrived before the ticket moves
  if (o == 0) {
    if (timed_out) {   // gave up waiting for a peer's chunk: raise the grid's sticky word BEFORE arriving, so that every block that passes the meeting point sees it
      __hip_atomic_store(oa.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(oa.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // bounded: if a block of the grid is not resident (a partitioned device, CUs held by another tenant — crl_ppo_create checks the
    // occupancy and turns the fused step off where the grid cannot fit, this is the second line of defence) or an arrival is lost, the
    // wait ends after oa.timeout ticks of the 100 MHz wall clock with a sticky error word instead of hanging the GPU; crl_sync /
    // crl_ppo_iterate / every read-back report it
    // The arrival count and the sticky word are neighbours (ticket[0], ticket[1]; the pair is 8-byte aligned): ONE 64-bit load per poll reads both, so the
    // load that lets a block through also tells it whether ANY block of the grid gave up before arriving — no second round trip on the step's critical path.
    const unsigned long long t0 = wall_clock64();
    unsigned long long both = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(oa.ticket), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((unsigned)both - oa.target > 0x7FFFFFFFu) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > oa.timeout) { __hip_atomic_store(oa.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); timed_out = 1; break; }
      both = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(oa.ticket), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((unsigned)(both >> 32) != 0u) timed_out = 1;
  }
  // If ANY block gave up (at the ticket, or waiting for a peer's chunk) NO block steps: a norm built from an incomplete chunk must not clip anybody's slice.
  // Parameters, Adam state and the β powers of the whole rank keep their pre-step values, the sticky words (ticket[1], and the peer exchange's own) report
  // the failure at the next host synchronisation, and — the word being sticky — every later launch on this handle skips its step too: the handle is poisoned
  // until it is destroyed, which is what a lost rank means for a data-parallel job anyway.
  if (__builtin_amdgcn_readfirstlane(timed_out)) return;
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  for (int a = a_lo; a <= a_hi; ++a) {
    const int b0 = oa.off[a] / 64, b1 = (oa.off[a + 1] - 1) / 64;
    double t = 0.0;
    for (int b = b0 + o; b <= b1; b += 64) t += __hip_atomic_load(oa.part + (size_t)a * gridDim.x + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = wave_sum(t);
    if (o == 0) nrm2[a] = t;
  }
  wave_lds_fence();
  if (!live) return;
  const float nrm = (float)sqrt(nrm2[arr]);
  const bool clip = (double)nrm > oa.thresh;
  const double sc = clip ? oa.thresh / (double)nrm : 1.0;
  const double b1c = 0.9, b2c = 0.999, epsn = 1e-8;
  double gd = (double)grad;
  if (clip) gd = (double)(float)(gd * sc);
  const float mi = (float)(b1c * (double)m_old + (1 - b1c) * gd);
  const float vi = (float)(b2c * (double)v_old + (1 - b2c) * gd * gd);
  oa.m[i] = mi; oa.v[i] = vi;
  const double delta = (double)mi / (1 - bp0) / (sqrt((double)vi / (1 - bp1)) + epsn) * oa.eta;
  oa.params[i] = p_old - (float)delta;
  if (i == oa.off[arr]) { oa.betap[2 * arr] = bp0 * b1c; oa.betap[2 * arr + 1] = bp1 * b2c; }
}

// Can the whole grid of reduce_optim_kernel be resident at once on this device? Its meeting point needs every block running: (P+4+63)/64
// blocks of 1024 threads (the gradient's chunks, the four loss sums riding in the last one: 144 for the 4/2/64 networks = 72 CUs at two blocks per CU). On a partitioned (CPX) device or a smaller GPU it
// may not fit; the handle then keeps the two-launch optimiser step (reduce_kernel + clipnorm_adam_kernel).
int fused_optim_fits(crl_ppo* h, bool* fits) {
  int per_cu = 0;
  CRL_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reduce_optim_kernel<true>, 64 * RG, 0));
  hipDeviceProp_t prop;
  CRL_HIP_CHECK(hipGetDeviceProperties(&prop, h->device));
  // one block per CU of margin: the occupancy query can read one high near a register-file edge (MI355X guide, "Residency")
  const long resident = (long)(per_cu > 1 ? per_cu - 1 : per_cu) * (long)prop.multiProcessorCount;
  *fits = resident >= (long)((h->P + 4 + 63) / 64);
  h->fuse_optim_capacity = (long)per_cu * (long)prop.multiProcessorCount;
  return 0;
}
// sticky time-out word of the meeting point (ticket[1]): read where the host synchronises anyway
int fused_optim_check(crl_ppo* h) {
  if (!h->ticket || h->ticket_target == 0) return 0;
  unsigned e = 0;
  CRL_HIP_CHECK(hipMemcpyAsync(&e, h->ticket + 1, sizeof(e), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (e) { set_error("reduce_optim_kernel: a block timed out at the grid meeting point (the grid was not fully resident, or the device is shared); set option fuse_optim = 0"); return 1; }
  return 0;
}

// data-parallel path: the sums are global only after the all-reduce, so the statistics get their own tiny launch
__global__ void stats_kernel(const float* __restrict__ msg, int P, StatsArgs st, int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (mode == 1 && st.vfix[3] == 0.0) return;
  compute_stats(msg, P, st.c, st.Mglobal, st.adv_ms, st.mb, st.vfix, st.out, mode);
}

// #{b : u > q_b} over the minibatch (only when the speculation flag is up)
__global__ void vfix_count_kernel(DevCfg c, const SampleRec* __restrict__ recs, const int32_t* __restrict__ perm, const float* __restrict__ newv,
                                  double* vfix) {
  if (vfix[3] == 0.0) return;
  __shared__ double sm[4];
  const float u = (float)vfix[0];
  double cnt = 0.0;
  for (int pos = threadIdx.x; pos < c.M; pos += blockDim.x) {
    const int idx = perm ? perm[pos] : pos;
    const float v = newv[pos], ov = recs[idx].old_v, R = recs[idx].ret;
    const float cl = fminf(fmaxf(v - ov, -c.clip), c.clip);
    const float vc = ov + cl;
    const float q = (vc - R) * (vc - R);
    cnt += (u > q) ? 1.0 : 0.0;
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sm[w];
    vfix[1] = s;
  }
}


// option update_tile: 32 = update_x2_kernel always; 16 (17: + the miss test hook) = the 16-sample-tile kernel always; 0 = automatic — the 16-sample kernel
// for launches with at most UPDATE16_MAX_TILES_PER_WAVE 32-sample tiles per wave of update_x2_kernel's grid (shards of 8192 envs and below, C2)
constexpr int UPDATE16_MAX_TILES_PER_WAVE = 8;
constexpr bool UPDATE16_AUTO = false;   // flipped only by a same-box A/B at 4096 / 8192 envs (profiles/r06_update_tile16_ab.txt)
static bool update_uses_tile16(const crl_ppo* h) {
  const int64_t t = opt(h, OPT_UPDATE_TILE);
  if (t == 16 || t == 17) return true;
  if (t == 32) return false;
  const long tiles32 = ((long)h->dc.M + 31) / 32;
  const long waves_per_role = (long)((h->update_blocks + 1) / 2) * 8;
  return UPDATE16_AUTO && tiles32 <= (long)UPDATE16_MAX_TILES_PER_WAVE * waves_per_role;
}

// block counts of the main pass: {actor, critic}
static void main_pass_blocks(crl_ppo* h, int* nA, int* nC) {
  // one role per block; the actor tile is a little longer (softmax + Float64 policy-loss terms), so it gets more blocks
  const int total = 2 * ((h->update_blocks + 1) / 2);
  const int actor_pct = (int)opt(h, OPT_ACTOR_BLOCK_PCT);  // swept 50..55 at nt=65536: 14.92 14.79 14.56 14.41 14.52 14.69 ms of update per iteration
  int a = total * actor_pct / 100;
  if (a < 1) a = 1;
  if (a > total - 1) a = total - 1;
  if (total < 2) { *nA = 1; *nC = 1; return; }
  // multiples of 8 where the grid allows: a role-local block index then names the block's XCD (UpdateArgs::xcd_align)
  if (total % 8 == 0 && total >= 32 && opt(h, OPT_UPDATE_XCD_ALIGN)) { a = ((a + 4) / 8) * 8; if (a < 8) a = 8; if (a > total - 8) a = total - 8; }
  *nA = a; *nC = total - a;
}

static int run_update(crl_ppo* h, int mb, int mode, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {
  UpdateArgs a;
  a.c = h->dc; a.params = h->params;
  a.recs = h->recs; a.perm = h->perm_base + (size_t)h->cur_slot * h->dc.B + (size_t)mb * h->dc.M; a.adv_ms = h->adv_ms; a.vfix = h->vfix;
  a.gpart = h->gpart; a.lpart = h->lpart; a.newv = h->newv; a.range_err = h->vfix + 5;
  a.dscale = h->dscale; a.dmax = reinterpret_cast<unsigned*>(h->dscale + 2); a.dw_miss = reinterpret_cast<unsigned*>(h->dscale + 4);
  a.mb = mb; a.mode = mode; a.gstride = (int)h->Pa; a.pmax = h->update_blocks; a.stagger = 0;
  a.Mglobal = (double)h->dc.M * h->world;
  if (mode == 1) {
    a.nblk[0] = 0; a.nblk[1] = h->update_blocks;
    const size_t smem = sizeof(float) * (NetImage<4, 1, true>::SIZE + 4 * SCR_FLOATS);
    hipLaunchKernelGGL((update_vfix_kernel<4, 2>), dim3(h->update_blocks), dim3(256), smem, h->stream, a);
  } else {
    main_pass_blocks(h, &a.nblk[0], &a.nblk[1]);
    a.xcd_align = (a.nblk[0] % 8 == 0 && a.nblk[1] % 8 == 0 && opt(h, OPT_UPDATE_XCD_ALIGN)) ? 1 : 0;
    a.stagger = (int)opt(h, OPT_UPDATE_STAGGER);
    a.prio_mode = (int)opt(h, OPT_UPDATE_PRIO_SMALL);
    // LDS for the larger of the two layouts: the fp16x2 kernel runs a role as bf16x3 when its weights leave the fp16 window
    const size_t smem = sizeof(float) * (X2_KERNEL_LDS_FLOATS + 8 * PF_SLOT_FLOATS);
    static_assert(NetImageX3<4, 2, true>::SIZE >= NetImageX2<4, 2>::SIZE, "the bf16x3 image is the larger one");
    if (gemm_x2(h) && update_uses_tile16(h)) {
      // small launches (option update_tile): 16-sample tiles, twelve waves per block = three per SIMD; a tile that does not fit the carried
      // weight-gradient scale raises a flag and the early-exit repair launch behind it redoes the minibatch as bf16x3
      const size_t smem16 = sizeof(float) * update16_smem_floats();
      a.stagger = 3;
      if (opt(h, OPT_UPDATE_TILE) == 17) a.mode = 2;   // test hook: every tile reports a miss, the repair launch produces the result
      hipExtLaunchKernelGGL((update_t16_kernel<2>), dim3(a.nblk[0] + a.nblk[1]), dim3(64 * RW16), smem16, h->stream, ev0, ev1, 0, a);
      a.mode = 0; a.stagger = (int)opt(h, OPT_UPDATE_STAGGER);
      hipLaunchKernelGGL((update_repair_kernel<4, 2>), dim3(a.nblk[0] + a.nblk[1]), dim3(512), smem, h->stream, a);
    } else if (gemm_x2(h)) hipExtLaunchKernelGGL((update_x2_kernel<4, 2>), dim3(a.nblk[0] + a.nblk[1]), dim3(512), smem, h->stream, ev0, ev1, 0, a);
    else hipExtLaunchKernelGGL((update_x3_kernel<4, 2>), dim3(a.nblk[0] + a.nblk[1]), dim3(512), smem, h->stream, ev0, ev1, 0, a);
  }
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

static void launch_vfix_count(crl_ppo* h, int mb) {
  const size_t off = (size_t)h->cur_slot * h->dc.B + (size_t)mb * h->dc.M;
  hipLaunchKernelGGL(vfix_count_kernel, dim3(1), dim3(1024), 0, h->stream, h->dc, h->recs, h->perm_base + off, h->newv, h->vfix);
}

static StatsArgs stats_args(crl_ppo* h, int mb, crl_ppo_stats* slot, int fused) {
  StatsArgs st;
  st.c = h->dc; st.Mglobal = (double)h->dc.M * h->world; st.adv_ms = h->adv_ms; st.mb = mb; st.vfix = h->vfix; st.out = slot;
  st.fused = fused;
  st.dscale = gemm_x2(h) ? h->dscale : nullptr;
  return st;
}

// One optimiser step's gradient: update pass → fixed-order reduce (+ statistics) → [all-reduce → statistics] →
// the rare exact value-loss pass (three early-exit launches). The gradient message ends up in comm_buf.
// inline_fix: follow the speculative pass with the three early-exit launches of the exact value-loss pass (host-driven single
// steps). crl_ppo_iterate passes false: there a failed speculation is caught by the guard window (api.cpp) instead.
int launch_update(crl_ppo* h, int mb, crl_ppo_stats* stats_slot, bool inline_fix, bool with_optim, double eta) {
  if (h->wide) return wide_update(h, mb, stats_slot);
  if (h->cfg.obs_dim != 4 || h->cfg.n_act != 2 || h->cfg.hidden != 64) {
    set_error("this build of libcleanrl_hip supports obs_dim=4, n_act=2, hidden=64 (2x64 MLP) only");
    return 1;
  }
  const int P = (int)h->P;
  const bool dp = has_comm(h);
  {
    ProfScope ps(h, CRL_K_UPDATE, /*attach=*/true);
    if (run_update(h, mb, 0, ps.a, ps.b)) return 1;
  }
  {
    int nA, nC;
    main_pass_blocks(h, &nA, &nC);
    ProfScope ps(h, CRL_K_REDUCE);
    if (with_optim) {
      // no fix-up between gradient and optimiser: reduce (+ the peer exchange under data parallelism) + ClipNorm + Adam in one launch
      if ((dp && !peer_active(h)) || (inline_fix && h->cfg.clip_value_loss) || (P & 63) > 60) {
        set_error("internal: the fused optimiser step needs a speculative step that is local or exchanged through the peer mailboxes"); return 1;
      }
      FusedOptimArgs oa;
      const int hN = h->cfg.hidden, d = h->cfg.obs_dim, A = h->cfg.n_act;
      const int sizes[12] = {hN * d, hN, hN * hN, hN, A * hN, A, hN * d, hN, hN * hN, hN, hN, 1};
      oa.off[0] = 0;
      for (int q = 0; q < 12; ++q) oa.off[q + 1] = oa.off[q] + sizes[q];
      const unsigned nb = (unsigned)((P + 4 + 63) / 64);     // the four loss sums ride behind the gradient, in the chunk of index P
      CRL_HIP_CHECK(hipGetLastError());   // an error left behind by an earlier call is reported as such, not mistaken for this launch's
      PeerArgs pa{};
      if (dp && peer_next_args(h, &pa, (int)nb, (size_t)P + 4)) return 1;
      h->ticket_target += nb;
      oa.params = h->params; oa.m = h->adam_m; oa.v = h->adam_v; oa.betap = h->betap; oa.part = h->optim_part; oa.ticket = h->ticket;
      oa.target = h->ticket_target; oa.eta = eta; oa.thresh = 0.5; oa.timeout = 200000000ull;
      if (dp) {
        // a peer that never arrives ends the wait after peer_timeout_ms; the grid's own meeting point must outlast that
        const unsigned long long pt = (unsigned long long)pa.timeout_ticks + 200000000ull;
        oa.timeout = pt;
        hipLaunchKernelGGL(reduce_optim_kernel<true>, dim3(nb), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, nA, nC, h->update_blocks, (int)h->Pa,
                           (int)h->Pa, (int)h->Pc, h->comm_buf, stats_args(h, mb, stats_slot, 1), oa, pa);
      } else {
        hipLaunchKernelGGL(reduce_optim_kernel<false>, dim3(nb), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, nA, nC, h->update_blocks, (int)h->Pa,
                           (int)h->Pa, (int)h->Pc, h->comm_buf, stats_args(h, mb, stats_slot, 1), oa, pa);
      }
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) {             // a launch that never ran must not leave later ones waiting for its arrivals
        h->ticket_target -= nb;
        set_error(std::string("reduce_optim_kernel launch failed: ") + hipGetErrorString(le));
        return 1;
      }
      wide_mark_params_changed(h);
    } else {
      hipLaunchKernelGGL(reduce_kernel<0>, dim3((P + 63) / 64), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, nA, nC,
                         h->update_blocks, (int)h->Pa, (int)h->Pa, (int)h->Pc, h->comm_buf, stats_args(h, mb, stats_slot, dp ? 0 : 1));
    }
    CRL_HIP_CHECK(hipGetLastError());
  }
  if (dp && !with_optim) {      // (the one-launch step has exchanged the message itself and written the statistics)
    {
      ProfScope ps(h, CRL_K_ALLREDUCE);
      if (comm_allreduce(h, h->comm_buf, (size_t)P + 4, false)) return 1;
    }
    if (h->defer_stats) { h->stats_pending = true; h->stats_mb = mb; h->stats_slot = stats_slot; }   // launch_optim carries them
    else hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(64), 0, h->stream, h->comm_buf, P, stats_args(h, mb, stats_slot, 0), 0);
    CRL_HIP_CHECK(hipGetLastError());
  }
  if (inline_fix && h->cfg.clip_value_loss && h->world == 1) {
    // early-exit launches unless the statistics raised the flag (u > 0)
    launch_vfix_count(h, mb);
    CRL_HIP_CHECK(hipGetLastError());
    if (run_update(h, mb, 1)) return 1;
    hipLaunchKernelGGL(reduce_kernel<1>, dim3((P + 63) / 64), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, 0, h->update_blocks,
                       h->update_blocks, (int)h->Pa, (int)h->Pa, (int)h->Pc, h->comm_buf, stats_args(h, mb, stats_slot, 1));
    CRL_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// Data-parallel step when the value-loss branch u > 0 may be live (ppo.jl:232-237, Q4): the speculative pass, its reduce and
// all-reduce run as usual; then the GLOBAL flag is read back (the same value on every rank — u is a global mean), and
// only if it is up do all ranks add: local #{u > q} → all-reduce → exact critic pass → its reduce → all-reduce of the
// critic slice and the two critic loss sums. Slow path by design (a host read-back per step): crl_ppo_iterate enters it
// only when re-running an iteration whose sticky flag was raised.
int launch_update_exact_dp(crl_ppo* h, int mb, crl_ppo_stats* stats_slot) {
  if (h->wide) return wide_update(h, mb, stats_slot);   // the layer-wise path is always exact
  const int P = (int)h->P, Pa = (int)h->Pa, Pc = (int)h->Pc;
  if (run_update(h, mb, 0)) return 1;
  int nA, nC;
  main_pass_blocks(h, &nA, &nC);
  hipLaunchKernelGGL(reduce_kernel<0>, dim3((P + 63) / 64), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, nA, nC, h->update_blocks,
                     Pa, Pa, Pc, h->comm_buf, stats_args(h, mb, stats_slot, 0));
  CRL_HIP_CHECK(hipGetLastError());
  if (comm_allreduce(h, h->comm_buf, (size_t)P + 4, false)) return 1;
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(64), 0, h->stream, h->comm_buf, P, stats_args(h, mb, stats_slot, 0), 0);
  CRL_HIP_CHECK(hipGetLastError());
  if (!h->cfg.clip_value_loss) return 0;
  double flag = 0.0;
  CRL_HIP_CHECK(hipMemcpyAsync(&flag, h->vfix + 3, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  CRL_HIP_CHECK(hipStreamSynchronize(h->stream));
  if (flag == 0.0) return 0;
  launch_vfix_count(h, mb);
  CRL_HIP_CHECK(hipGetLastError());
  if (comm_allreduce(h, h->vfix + 1, 1, true)) return 1;                       // global #{u > q}
  if (run_update(h, mb, 1)) return 1;                                          // exact critic gradient of this shard
  hipLaunchKernelGGL(reduce_kernel<1>, dim3((P + 63) / 64), dim3(64 * RG), 0, h->stream, h->gpart, h->lpart, 0, h->update_blocks,
                     h->update_blocks, Pa, Pa, Pc, h->comm_buf, stats_args(h, mb, stats_slot, 0));
  CRL_HIP_CHECK(hipGetLastError());
  if (comm_allreduce(h, h->comm_buf + Pa, (size_t)Pc, false)) return 1;        // critic slice (the actor slice is global already)
  if (comm_allreduce(h, h->comm_buf + P + 2, 2, false)) return 1;              // Σ(v − R²), Σ max(u, q)
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(64), 0, h->stream, h->comm_buf, P, stats_args(h, mb, stats_slot, 0), 1);
  CRL_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace crl

#ifdef CRL_EXP_STAMPS
extern "C" int32_t crl_debug_read_tstamps(unsigned* out, int32_t n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(crl::crl_dbg_tstamps), sizeof(unsigned) * (size_t)n) == hipSuccess ? 0 : 1;
}
extern "C" int32_t crl_debug_read_stamps(unsigned long long* out, int32_t n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(crl::crl_dbg_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 1;
}
#endif
