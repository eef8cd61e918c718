"""Gradient error of the 2x256 update pass on a LATE-TRAINING state (CartPole, 2x256, after N PPO iterations): the state is produced with round 5's kernels,
then every wide_rs flavour and the oracle differentiate the same minibatches.   python scripts/diag_rs_late.py [iterations]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cleanrl_jl_amd as crl
import oraclelib as O
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
nt, k, D, A, Hd = 1024, 128, 4, 2, 256
L = crl._lib
pc = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 200)
agent = crl.Agent(pc, obs_dim=D, n_act=A, hidden=Hd, env_kind=L.ENV_CARTPOLE, seed=1, init_seed=1, options={"wide_rs": 0})
h = agent.handle; F = L
h.env_reset()
for _ in range(iters): h.iterate(1, want_stats=False)
h.rollout_run(); h.compute_gae()
params = h.read(F.F_PARAMS)
bufs = {f: h.read(f) for f in (F.F_OBS, F.F_ACTION, F.F_LOGPROB, F.F_VALUE, F.F_ADVANTAGE, F.F_RETURN, F.F_PERM)}
es = h.episode_stats(); print("state after %d iterations: mean episode return of the last rollout %.1f; |params| max %.2f; |value| max %.1f" % (iters, es["return_sum"] / max(es["episodes"], 1), np.abs(params).max(), np.abs(bufs[F.F_VALUE]).max()))
agent.close()
cfg = O.make_config(num_envs=nt, num_steps=k, obs_dim=D, n_act=A, hidden=Hd, env_kind=0)
off = O.param_offsets(cfg)
M = nt * k // 4
names = ["aW1", "ab1", "aW2", "ab2", "aW3", "ab3", "cW1", "cb1", "cW2", "cb2", "cW3", "cb3"]
perm = bufs[F.F_PERM]
orc = {}
for mb in (0, 2):
    orc[mb] = O.loss_grad(cfg, params, bufs[F.F_OBS].reshape(D, -1, order="F"), bufs[F.F_ACTION], bufs[F.F_LOGPROB], bufs[F.F_VALUE], bufs[F.F_ADVANTAGE], bufs[F.F_RETURN], perm[mb * M:(mb + 1) * M])
for rs in (0, 1, 8, 27):
    a2 = crl.Agent(pc, params=params, obs_dim=D, n_act=A, hidden=Hd, env_kind=L.ENV_CARTPOLE, options={"wide_rs": rs})
    h2 = a2.handle
    for f, a in bufs.items(): h2.write(f, a)
    h2.adv_stats()
    for mb in (0, 2):
        gs = h2.update_minibatch(mb, 2.5e-4, apply_update=False)
        g = h2.read(F.F_GRADS).astype(np.float64); go, so = orc[mb]
        errs = [np.linalg.norm(g[off[i]:off[i + 1]] - go[off[i]:off[i + 1]]) / max(np.linalg.norm(go[off[i]:off[i + 1]]), 1e-30) for i in range(12)]
        print("wide_rs=%2d mb %d  loss %.7e (oracle %.7e) v_loss %.6e (%.6e)  rel-L2: %s" % (rs, mb, gs["loss"], so["loss"], gs["v_loss"], so["v_loss"], " ".join("%s %.1e" % (n, e) for n, e in zip(names, errs))))
    a2.close()
