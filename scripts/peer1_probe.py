import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import cleanrl_jl_amd as crl
L = crl._lib
res = {}
for name, peer, fuse in (("local_fused", 0, 1), ("local_two", 0, 0), ("peer_fused", 1, 1), ("peer_three", 1, 0)):
    a = crl.Agent(crl.PPOConfig(num_envs=256, num_steps=128, total_timesteps=256 * 128 * 10), init_seed=4, options={"fuse_optim": fuse})
    h = a.handle
    if peer:
        h.comm_peer_attach(h.comm_peer_export(1, 0))
    h.env_reset()
    st = h.iterate(1)
    res[name] = (st[0]["loss"], st[-1]["loss"], st[-1]["v_loss"], float(np.linalg.norm(h.read(L.F_PARAMS).astype(np.float64))), float(np.linalg.norm(h.read(L.F_GRADS).astype(np.float64))))
    a.close()
for k, v in res.items():
    print(k, v)
