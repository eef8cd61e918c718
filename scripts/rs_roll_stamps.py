"""In-kernel timeline of one step (step 64) of wide_rs_rollout_kernel (diagnostic build: bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide).
CRL_LIB_PATH=cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so python scripts/rs_roll_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402

L = crl._lib
cfg = crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100)
agent = crl.Agent(cfg, obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC, options={"wide_rs": 2})
h = agent.handle
h.env_reset(); h.rollout_run(); h.sync()
lib = L.load()
buf = np.zeros(256 * 8 * 16, np.uint64)
lib.crl_debug_read_rs_stamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_rs_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
s = buf.reshape(256, 8, 16).astype(np.int64)
d = lambda a, b, w: np.median((s[:, w, b] - s[:, w, a]) / 100.0)
for w, nm in ((slice(0, 1), "wave 0"), (slice(1, 8), "waves 1-7")):
    print("%-10s P(0) + L1(1) %.2f | B1 %.2f | M(0) || P(1) %.2f | B2 %.2f | M(1) || E(0), E(1) %.2f | B3 %.2f | S %.2f | B4 %.2f | step %.2f us" % (
        nm, d(0, 1, w), d(1, 2, w), d(2, 3, w), d(3, 4, w), d(4, 5, w), d(5, 6, w), d(6, 7, w), d(7, 9, w), d(0, 9, w)))
print("wave 0: fold %.2f | env step %.2f" % (d(6, 8, slice(0, 1)), d(8, 7, slice(0, 1))))
agent.close()
