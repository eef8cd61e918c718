"""Timeline of one slab (the 41st of every block's chunk) of wide_wgrad_gen_kernel (build: bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402
L = crl._lib
agent = crl.Agent(crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100), obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
h = agent.handle; h.env_reset(); h.iterate(1, want_stats=False); h.sync()
lib = L.load(); buf = np.zeros(2 * 256 * 8 * 16, np.uint64)
lib.crl_debug_read_wstamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_wstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(2, 256, 8, 16)[0, :128, :, :8].astype(np.int64)
ok = (st[:, :, 0] > 0).all(axis=1); st = st[ok]
us = (st - st[:, :, :1].min(axis=1, keepdims=True)) / 100.0
names = ["slab top", "after barrier 1", "delta2 split + LDS stores issued (split kernel: wave 0's side requests issued)", "h1 tile made (MFMA, tanh, split, stores)", "after barrier 2", "next slab's loads issued (split kernel: the plane DMA of the slab after next)",
         "first k-step's 24 MFMAs issued", "second k-step issued (slab end)"]
print(len(st), "blocks; us since the first wave of the block reached the top of its 41st slab (median over blocks and waves; min .. max over waves of the block medians)")
for i, n in enumerate(names):
    print("  %-45s %5.2f   (%.2f .. %.2f)" % (n, np.median(us[..., i]), np.median(us[..., i].min(axis=1)), np.median(us[..., i].max(axis=1))))
agent.close()
