"""In-kernel timeline of one stage (stage 20 of every actor block) of the register-stationary forward, wide_rs_fwd_kernel (diagnostic build:
bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide).   CRL_LIB_PATH=cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so python scripts/rs_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402

L = crl._lib
cfg = crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100)
agent = crl.Agent(cfg, obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC, options={"wide_rs": 1})
h = agent.handle
h.env_reset(); h.iterate(1, want_stats=False); h.sync()
lib = L.load()
buf = np.zeros(256 * 8 * 16, np.uint64)
lib.crl_debug_read_rs_stamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_rs_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
s = buf.reshape(256, 8, 16)[:128].astype(np.int64)
d = lambda a, b: np.median((s[..., b] - s[..., a]) / 100.0)
print("stage 20, median over the actor's blocks and waves (us): fold + layer 1 of the next tile %.2f | group 0 %.2f | group 1 %.2f | group 2 %.2f | group 3 %.2f | "
      "head sums + accumulator hand-over %.2f | wait + barrier %.2f | stage %.2f" % (d(0, 1), d(1, 5), d(5, 6), d(6, 7), d(7, 8), d(2, 3), d(3, 4), d(0, 4)))
agent.close()
