"""GAE kernel micro-benchmark: avg launch time / GB/s over the resident rollout buffer (17 B per (env,step) + 5 B per env)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl
for nt in (4096, 8192, 16384, 65536):
    agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=128))
    h = agent.handle
    h.env_reset(); h.rollout_run()
    for _ in range(5): h.compute_gae()
    h.sync(); h.prof_enable(True); h.prof_reset()
    for _ in range(50): h.compute_gae()
    ms, n = h.prof_read()["gae"]
    b = 17 * nt * 128 + 5 * nt
    print(f"L={os.environ.get('CRL_GAE_L','-')} EB={os.environ.get('CRL_GAE_EB','-')} nt={nt}: {ms/n*1e3:.2f} us  {b/(ms/n*1e-3)/1e9:.0f} GB/s  ({b/(ms/n*1e-3)/8e12*100:.1f}% of 8 TB/s)")
    agent.close()
