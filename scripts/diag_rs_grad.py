"""Gradient error of the 2x256 update pass against the oracle for several wide_rs flavours at a training-sized minibatch (diagnostic).
   python scripts/diag_rs_grad.py [nt k scale]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CRL_FORCE_WIDE"] = "1"
import cleanrl_jl_amd as crl
import oraclelib as O
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
D, A, Hd = 4, 2, 256
cfg = O.make_config(num_envs=nt, num_steps=k, obs_dim=D, n_act=A, hidden=Hd, env_kind=1)
rng = np.random.default_rng(3)
params = (O.orthogonal_params(cfg, 5) + (0.05 * rng.standard_normal(O.lib().orc_param_count(cfg))).astype(np.float32)) * np.float32(scale)
off = O.param_offsets(cfg)
st = O.State(cfg); st.params[:] = params
st.obs[:] = (rng.standard_normal((D, nt, k)) * np.array([1.0, 2.0, 0.1, 2.5])[:, None, None]).astype(np.float32)
st.action[:] = rng.integers(0, A, (nt, k)); st.logprob[:] = (np.log(1.0 / A) + 0.3 * rng.standard_normal((nt, k))).astype(np.float32)
st.value[:] = (50 * rng.standard_normal((nt, k))).astype(np.float32); st.adv[:] = (2 * rng.standard_normal((nt, k))).astype(np.float32)
st.ret[:] = (60 * rng.standard_normal((nt, k))).astype(np.float32); st.perm[:] = rng.permutation(nt * k).astype(np.int32)
M = nt * k // 4
g_orc, so = O.loss_grad(cfg, params, st.obs.reshape(D, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, st.perm[:M])
names = ["aW1", "ab1", "aW2", "ab2", "aW3", "ab3", "cW1", "cb1", "cW2", "cb2", "cW3", "cb3"]
for rs in (0, 1, 8, 24, 27):
    pc = crl.PPOConfig(num_envs=nt, num_steps=k, total_timesteps=nt * k * 10)
    agent = crl.Agent(pc, params=params, obs_dim=D, n_act=A, hidden=Hd, env_kind=crl._lib.ENV_SYNTHETIC, options={"wide_rs": rs})
    h = agent.handle; F = crl._lib
    for f, a in ((F.F_OBS, st.obs), (F.F_ACTION, st.action), (F.F_LOGPROB, st.logprob), (F.F_VALUE, st.value), (F.F_ADVANTAGE, st.adv), (F.F_RETURN, st.ret), (F.F_PERM, st.perm)):
        h.write(f, a)
    h.adv_stats()
    gs = h.update_minibatch(0, 2.5e-4, apply_update=False)
    g = h.read(F.F_GRADS).astype(np.float64)
    errs = [np.linalg.norm(g[off[i]:off[i + 1]] - g_orc[off[i]:off[i + 1]]) / max(np.linalg.norm(g_orc[off[i]:off[i + 1]]), 1e-30) for i in range(12)]
    print("wide_rs=%2d  loss %.6e (oracle %.6e)  rel-L2 per array: %s" % (rs, gs["loss"], so["loss"], " ".join("%s %.1e" % (n, e) for n, e in zip(names, errs))))
    agent.close()
st.close()
