/*
 * dqn_oracle.h — CPU restatement of the reference DQN loop (TEST INFRASTRUCTURE ONLY).
 *
 * Restates sash-a/CleanRL.jl `src/algorithms/dqn.jl` in plain C. PARITY UNPINNED (no reference tests / fixtures, no Julia
 * here; third-party arithmetic marked [3P-memory]). Numeric regime as in a2c_oracle.h: `CartPoleEnv()` is Float64, the
 * Float32-weight Dense layers promote, so Q values, TD targets, the loss and every cotangent are Float64; gradients are
 * projected to Float32 and Adam state is Float32.
 * Random choices (ε draw, random action, the minibatch drawn without replacement) come from this build's Philox streams;
 * the sampler is a self-avoiding draw (uniform over k-permutations like StatsBase.sample(1:n, k; replace=false)).
 *
 * Only tests/ may load this library.
 */
#ifndef DQN_ORACLE_H
#define DQN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {                 /* DQNConfig, dqn.jl:1-19 */
  int64_t log_frequency, total_timesteps, buffer_size, min_buff_size;
  double lr;
  int64_t train_freq, target_net_freq, batch_size;
  double gamma, epsilon_start, epsilon_end, epsilon_duration;
  int32_t max_steps;             /* CartPoleEnv() default 200 (dqn.jl:37) */
  int32_t pad;
  uint64_t seed;
} dqn_config;

#define DQN_H1 120
#define DQN_H2 84
#define DQN_D 4
#define DQN_A 2
#define DQN_P (DQN_H1 * DQN_D + DQN_H1 + DQN_H2 * DQN_H1 + DQN_H2 + DQN_A * DQN_H2 + DQN_A)   /* 10,934 */

typedef struct { double episode_return; int64_t episode_length, global_step; double epsilon; } dqn_episode;  /* dqn.jl:88 */
typedef struct { int64_t global_step; double loss; } dqn_loss_record;                                          /* dqn.jl:116 */

typedef struct dqn_state dqn_state;
dqn_state* dqn_create(const dqn_config* c);     /* dqn.jl:34-56 */
void dqn_destroy(dqn_state* s);
void dqn_set_params(dqn_state* s, const float* q_params);   /* q_net; target_net = deepcopy (dqn.jl:40) */
void dqn_get_params(const dqn_state* s, float* q_params, float* target_params);
void dqn_get_env(const dqn_state* s, double* state4, int64_t* global_step, int64_t* rb_size, double* last_loss, int64_t* n_updates);
/* dqn.jl:57-119: runs up to max_env_steps iterations of the loop (or to total_timesteps). Returns steps taken. */
int64_t dqn_run(dqn_state* s, int64_t max_env_steps, dqn_episode* eps, int32_t max_eps, int32_t* n_eps,
                dqn_loss_record* losses, int32_t max_losses, int32_t* n_losses);

/* pieces for unit tests */
double dqn_linear_schedule(double start_e, double end_e, double duration, double t);                /* dqn.jl:28-31 */
void dqn_forward(const float* params, const double* x, double* q);                                  /* make_nn, dqn.jl:22-26 */
/* dqn.jl:96-108: TD target from the target net, mse loss, Float32-projected gradient of the q net */
double dqn_loss_grads(const float* q_params, const float* target_params, const double* state, const double* next_state,
                      const int32_t* action, const double* reward, const uint8_t* terminal, int32_t n, double gamma, float* grads);
/* the build's sampler: k distinct indices of [0, n) in draw order */
void dqn_sample_indices(uint64_t seed, uint64_t gstep, int32_t n, int32_t k, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
