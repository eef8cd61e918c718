#!/usr/bin/env python3
"""A2C side measurement: env-steps/s of the whole a2c.jl loop (single CartPoleEnv{Float64}, collect +
updates) on the GPU vs the CPU restatement of the reference (oracle, one host thread — the reference loop is serial)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import cleanrl_jl_amd as crl
import oraclelib as O

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
params = O.orthogonal_params(O.make_config(), 1)
agent = crl.A2CAgent(crl.A2CConfig(total_timesteps=T), params=params, seed=1)
h = agent.handle
h.run_until_update(max_env_steps=2000)   # warm-up
t0 = time.perf_counter(); s0 = h.env()[1]; upd = 0
while h.env()[1] < T:
    taken, ts, eps = h.run_until_update()
    upd += ts["trained"]
    if taken == 0: break
dt_g = time.perf_counter() - t0; steps_g = h.env()[1] - s0
st = O.A2CState(O.a2c_config(total_timesteps=T, seed=1), params)
st.run_until_update(max_env_steps=2000)
t0 = time.perf_counter(); s0 = st.env()[1]; budget = 15.0
while st.env()[1] < T and time.perf_counter() - t0 < budget:
    taken, ts, eps = st.run_until_update()
    if taken == 0: break
dt_c = time.perf_counter() - t0; steps_c = st.env()[1] - s0
print(json.dumps({"metric": "A2C env-steps/s (single CartPoleEnv{Float64}, a2c.jl loop incl. updates)",
                  "gpu": {"steps": steps_g, "seconds": dt_g, "steps_per_s": steps_g / dt_g, "updates": upd},
                  "cpu_oracle_1_thread": {"steps": steps_c, "seconds": dt_c, "steps_per_s": steps_c / dt_c}}))
