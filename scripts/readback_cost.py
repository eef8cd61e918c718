#!/usr/bin/env python3
"""Where the read-back loop's extra time goes (bench.py `with_stats_readback` vs the headline): per-iteration ms of
(a) enqueue only, (b) + crl_sync per iteration, (c) + the 16 loss records, (d) + the episode statistics (what ppo() / train() did up to round 5), (e) the same records through crl_ppo_iterate_async (what they do now)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = crl.PPOConfig(num_envs=nt, num_steps=128, total_timesteps=nt * 128 * 400)
agent = crl.Agent(cfg)
h = agent.handle
h.env_reset()
for _ in range(5):
    h.iterate(1, want_stats=False)
h.sync()
out = {}
def run(name, fn, n=30):
    h.sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    h.sync(); out[name] = (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    run("a_enqueue_only", lambda: h.iterate(1, want_stats=False))
    run("b_sync_each", lambda: (h.iterate(1, want_stats=False), h.sync()))
    run("c_stats_each", lambda: h.iterate(1, want_stats=True))
    run("d_stats_and_episode_stats", lambda: (h.iterate(1, want_stats=True), h.episode_stats()))
    run("e_pipelined_iterate_async", lambda: h.iterate_async(want_stats=True)); h.drain()
print(json.dumps({"num_envs": nt, "ms_per_iteration": out}))
agent.close()
