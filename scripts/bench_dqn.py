#!/usr/bin/env python3
"""DQN side measurement: env-steps/s of the whole dqn.jl loop (single CartPoleEnv{Float64}, ε-greedy
collection + a 120-sample update every 10 steps) on the GPU vs the CPU restatement (oracle, one host thread)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cleanrl_jl_amd as crl
import oraclelib as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
params = O.dqn_params(1)
agent = crl.DQNAgent(crl.DQNConfig(total_timesteps=T), params=params, seed=1)
h = agent.handle
h.run(2000)
t0 = time.perf_counter(); s0 = h.status()["global_step"]
while h.status()["global_step"] < T:
    if h.run(20_000)[0] == 0: break
dt_g = time.perf_counter() - t0; steps_g = h.status()["global_step"] - s0
st = O.DQNState(O.dqn_config(total_timesteps=T, seed=1), params)
st.run(2000)
t0 = time.perf_counter(); s0 = st.env()["global_step"]
while st.env()["global_step"] < T and time.perf_counter() - t0 < 15.0:
    if st.run(5000)[0] == 0: break
dt_c = time.perf_counter() - t0; steps_c = st.env()["global_step"] - s0
print(json.dumps({"metric": "DQN env-steps/s (single CartPoleEnv{Float64}, dqn.jl loop incl. one 120-sample update per 10 steps)",
                  "gpu": {"steps": steps_g, "seconds": dt_g, "steps_per_s": steps_g / dt_g, "updates": h.status()["n_updates"]},
                  "cpu_oracle_1_thread": {"steps": steps_c, "seconds": dt_c, "steps_per_s": steps_c / dt_c}}))
