// init.cpp — host-side network initialisers: what `Networks.make_actor_critic` (networks.jl:36-49, called at ppo.jl:87 and a2c.jl:37) and
// `make_nn` (dqn.jl:22-26) leave in the Flux layers, as ONE flat float vector in Flux.params order. Run once per handle, on the host:
// 9,155 / 137,477 / 10,934 numbers — nothing for a kernel to win. They exist so that NO entry point of the boundary can train the all-zero
// network a fresh handle holds: a caller either uploads the reference's own `Flux.params` (crl_ppo_write(CRL_F_PARAMS)) or asks for this
// reference-shaped start; anything that computes with unset parameters is an error (api.cpp: params_set).
//
// [3P-memory, Flux 0.13.4] `Flux.orthogonal(rng, rows, cols; gain)`: rows < cols → the transpose of orthogonal(cols, rows);
// otherwise mat = randn(Float32, rows, cols); Q, R = qr(mat); Q · sign.(Diagonal(R)) · gain. `Dense(in => out; init)` has a zero bias.
// `Flux.glorot_uniform(out, in)` = (rand(Float32, out, in) .- 0.5) .* 2·sqrt(6 / (in + out)) — the default init of the DQN layers.
// The random STREAM is this library's own (splitmix64 + Box–Muller): Flux draws from Julia's task-local Xoshiro, which nobody outside
// Julia reproduces; the DISTRIBUTION (Haar-orthogonal columns times the gain, zero biases) is the reference's.
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/cleanrl_hip.h"

namespace crl {
void set_error(const std::string& msg);

namespace {
struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
  double normal() {
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

// W (rows, cols) column-major = gain · (Haar-orthogonal): Gaussian (n, m) with n = max, m = min, columns orthonormalised by modified
// Gram–Schmidt run twice (the second pass removes what rounding left) — the Q of the QR factorisation whose R has a positive diagonal,
// i.e. Householder's Q · sign(diag R) — transposed when rows < cols.
void orthogonal(Rng& rng, int rows, int cols, double gain, float* out) {
  const int n = rows > cols ? rows : cols, m = rows > cols ? cols : rows;
  std::vector<double> a((size_t)n * m);
  for (double& x : a) x = rng.normal();
  for (int j = 0; j < m; ++j) {
    double* cj = &a[(size_t)j * n];
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < j; ++i) {
        const double* ci = &a[(size_t)i * n];
        double d = 0.0;
        for (int r = 0; r < n; ++r) d += ci[r] * cj[r];
        for (int r = 0; r < n; ++r) cj[r] -= d * ci[r];
      }
    double nn = 0.0;
    for (int r = 0; r < n; ++r) nn += cj[r] * cj[r];
    nn = std::sqrt(nn);
    for (int r = 0; r < n; ++r) cj[r] /= nn;
  }
  for (int c = 0; c < cols; ++c)
    for (int r = 0; r < rows; ++r)
      out[(size_t)c * rows + r] = (float)(gain * (rows >= cols ? a[(size_t)c * n + r] : a[(size_t)r * n + c]));
}

void glorot_uniform(Rng& rng, int out_dim, int in_dim, float* out) {
  const double s = 2.0 * std::sqrt(6.0 / (double)(in_dim + out_dim));
  for (size_t i = 0; i < (size_t)out_dim * in_dim; ++i) out[i] = (float)((rng.uniform() - 0.5) * s);
}
}  // namespace
}  // namespace crl

using crl::set_error;

extern "C" {

int32_t crl_make_actor_critic(int32_t obs_dim, int32_t n_act, int32_t hidden, uint64_t seed, float* out, size_t n) {
  if (obs_dim < 1 || n_act < 1 || hidden < 1 || !out) { set_error("crl_make_actor_critic: bad arguments"); return 1; }
  const size_t h = (size_t)hidden, d = (size_t)obs_dim, A = (size_t)n_act;
  const size_t want = 2 * (h * d + h + h * h + h) + A * h + A + h + 1;
  if (n != want) { set_error("crl_make_actor_critic: expected " + std::to_string(want) + " floats, got " + std::to_string(n)); return 1; }
  crl::Rng rng{seed ^ 0xC1EA9B1ull};
  const double g = std::sqrt(2.0);
  float* p = out;
  for (size_t i = 0; i < n; ++i) out[i] = 0.0f;                                  // biases: zeros (Dense's default)
  // actor: mlp([in, h, h]) gain √2 (networks.jl:6-13,45), final layer gain 0.01 (networks.jl:40,42)
  crl::orthogonal(rng, hidden, obs_dim, g, p); p += h * d + h;
  crl::orthogonal(rng, hidden, hidden, g, p); p += h * h + h;
  crl::orthogonal(rng, n_act, hidden, 0.01, p); p += A * h + A;
  // critic: same trunk shape, final layer gain 1.0 (networks.jl:41,43,46)
  crl::orthogonal(rng, hidden, obs_dim, g, p); p += h * d + h;
  crl::orthogonal(rng, hidden, hidden, g, p); p += h * h + h;
  crl::orthogonal(rng, 1, hidden, 1.0, p);
  return 0;
}

int32_t crl_dqn_make_nn(uint64_t seed, float* out, size_t n) {
  if (!out || n != (size_t)CRL_DQN_PARAM_COUNT) { set_error("crl_dqn_make_nn: expected 10934 floats"); return 1; }
  crl::Rng rng{seed ^ 0xD09ull};
  for (size_t i = 0; i < n; ++i) out[i] = 0.0f;
  float* p = out;
  crl::glorot_uniform(rng, 120, 4, p); p += 120 * 4 + 120;     // dqn.jl:25 Dense(4, 120, relu)
  crl::glorot_uniform(rng, 84, 120, p); p += 84 * 120 + 84;    //           Dense(120, 84, relu)
  crl::glorot_uniform(rng, 2, 84, p);                          //           Dense(84, 2)
  return 0;
}

}  // extern "C"
