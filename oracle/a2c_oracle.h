/*
 * a2c_oracle.h — CPU restatement of the reference A2C loop (TEST INFRASTRUCTURE ONLY).
 *
 * Restates sash-a/CleanRL.jl `src/algorithms/a2c.jl` in plain C. PARITY UNPINNED for the same reasons as
 * ppo_oracle.h (no reference tests or fixtures, no Julia in the image, third-party arithmetic marked [3P-memory]).
 *
 * Numeric regime (a2c.jl:35,54-56): `CartPoleEnv(max_steps=500)` keeps its default element type Float64, so the
 * observation is a Vector{Float64}; Flux 0.13.4 Dense(W::Matrix{Float32}) applied to it promotes: every activation,
 * probability, loss and cotangent is Float64, the weights / Adam state / projected gradients are Float32.
 * (The Float64 method signature of discounted_future_rewards, a2c.jl:13, only type-checks in that regime.)
 *
 * Only tests/ may load this library.
 */
#ifndef A2C_ORACLE_H
#define A2C_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  double lr;                 /* a2c.jl:4  */
  int64_t total_timesteps;   /* a2c.jl:6  */
  int32_t min_replay_size;   /* a2c.jl:7  */
  double gamma;              /* a2c.jl:9  */
  int32_t obs_dim, n_act, hidden;   /* CartPole: 4, 2; networks.jl:36 default 64 */
  int32_t max_steps;         /* a2c.jl:35 max_steps=500 */
  uint64_t seed;
} a2c_config;

typedef struct { double actor_loss, critic_loss; int32_t n; int32_t trained; } a2c_train_stats;  /* a2c.jl:100 */
typedef struct { double episode_return; int64_t episode_length, global_step; } a2c_episode;      /* a2c.jl:106 */

typedef struct a2c_state a2c_state;
a2c_state* a2c_create(const a2c_config* c);   /* a2c.jl:32-52: buffer of 2*min_replay_size, reset!(env) */
void a2c_destroy(a2c_state* s);
int32_t a2c_param_count(const a2c_state* s);
void a2c_set_params(a2c_state* s, const float* p);
void a2c_get_params(const a2c_state* s, float* p);
void a2c_get_env(const a2c_state* s, double* state4, int64_t* global_step, int32_t* rb_size);
/* rollout buffer columns 1..size (a2c.jl:77): state (obs_dim,size) f64, action 0-based, reward, terminal */
void a2c_get_buffer(const a2c_state* s, double* state, int32_t* action, double* reward, uint8_t* terminal);

/* a2c.jl:53-111: steps the loop until ONE training update has run (ts->trained = 1) or max_env_steps were taken or
 * total_timesteps is reached. Episode records go to eps[0..*n_eps) (at most max_eps). Returns env steps taken. */
int64_t a2c_run_until_update(a2c_state* s, int64_t max_env_steps, a2c_train_stats* ts, a2c_episode* eps, int32_t max_eps,
                             int32_t* n_eps);

/* pieces, for unit tests */
void a2c_discounted_future_rewards(const double* rewards, const uint8_t* terminals, int32_t n, double final_value,
                                   double gamma, double* out);                                   /* a2c.jl:13-24 */
double a2c_tanh_fast(double x);                                                                   /* NNlib tanh_fast(::Float64) */
double a2c_sin(double x);
double a2c_cos(double x);
void a2c_forward(const a2c_config* c, const float* params, int net, const double* x, double* out); /* Float64 activations */
/* a2c.jl:81-97: losses and Float32-projected gradients of one training batch (critic arrays 6..11, actor arrays 0..5) */
void a2c_loss_grads(const a2c_config* c, const float* params, const double* states, const int32_t* actions,
                    const double* returns, int32_t n, float* grads, double* critic_loss, double* actor_loss, double* advantage);
void a2c_cartpole_step(double* s, int32_t* t, int32_t action, int32_t max_steps, int32_t* done);

#ifdef __cplusplus
}
#endif
#endif
