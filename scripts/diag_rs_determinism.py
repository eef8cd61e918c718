"""Bitwise repeatability of the 2x256 update pass per wide_rs flavour (diagnostic): the same minibatch N times, gradients compared bit for bit.
   python scripts/diag_rs_determinism.py [nt k repeats]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cleanrl_jl_amd as crl
import oraclelib as O
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
D, A, Hd = 4, 2, 256
cfg = O.make_config(num_envs=nt, num_steps=k, obs_dim=D, n_act=A, hidden=Hd, env_kind=1)
rng = np.random.default_rng(3)
params = O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)
obs = (rng.standard_normal((D, nt, k)) * np.array([1.0, 2.0, 0.1, 2.5])[:, None, None]).astype(np.float32)
act = rng.integers(0, A, (nt, k)).astype(np.int32); lp = (np.log(1.0 / A) + 0.3 * rng.standard_normal((nt, k))).astype(np.float32)
val = (50 * rng.standard_normal((nt, k))).astype(np.float32); adv = (2 * rng.standard_normal((nt, k))).astype(np.float32)
ret = (60 * rng.standard_normal((nt, k))).astype(np.float32); perm = rng.permutation(nt * k).astype(np.int32)
for rs in (0, 1, 8, 24, 27):
    pc = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10)
    agent = crl.Agent(pc, params=params, obs_dim=D, n_act=A, hidden=Hd, env_kind=crl._lib.ENV_SYNTHETIC, options={"wide_rs": rs})
    h = agent.handle; F = crl._lib
    for f, a in ((F.F_OBS, obs), (F.F_ACTION, act), (F.F_LOGPROB, lp), (F.F_VALUE, val), (F.F_ADVANTAGE, adv), (F.F_RETURN, ret), (F.F_PERM, perm)):
        h.write(f, a)
    h.adv_stats()
    ref = None; bad = 0; worst = 0.0
    for r in range(reps):
        h.update_minibatch(r % 4, 2.5e-4, apply_update=False)
        g = h.read(F.F_GRADS)
        if r < 4:
            ref = ref or {}; ref[r % 4] = g.copy()
        else:
            d = np.abs(g.astype(np.float64) - ref[r % 4].astype(np.float64))
            if not np.array_equal(g, ref[r % 4]):
                bad += 1; worst = max(worst, float(d.max() / max(np.abs(ref[r % 4]).max(), 1e-30)))
    print("wide_rs=%2d: %d of %d repeats differ from the first pass bit for bit (largest difference %.2e of the largest gradient entry)" % (rs, bad, reps - 4, worst))
    agent.close()
