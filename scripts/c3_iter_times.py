"""Per-iteration wall times of the C3 workload (one sync per iteration): python scripts/c3_iter_times.py <wide_rs> [iterations]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402
L = crl._lib
rs = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 1000)
agent = crl.Agent(cfg, obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC, options={"wide_rs": rs})
h = agent.handle
h.env_reset()
for _ in range(3):
    h.iterate(1, want_stats=False)
h.sync()
ts = []
for _ in range(n):
    t0 = time.perf_counter(); h.iterate(1, want_stats=False); h.sync(); ts.append((time.perf_counter() - t0) * 1e3)
s = sorted(ts)
print("wide_rs=%d: median %.2f min %.2f max %.2f ms; in order: %s" % (rs, s[len(s) // 2], s[0], s[-1], " ".join("%.1f" % t for t in ts)))
agent.close()
