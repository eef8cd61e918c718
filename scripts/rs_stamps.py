"""In-kernel timeline of the register-stationary forward (diagnostic build: bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide).
CRL_LIB_PATH=cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so python scripts/rs_stamps.py
Phases 16 / 17 of every actor block; M-phase slots: 0 top, 1 products issued, 2 layer-1 staged + sum, 3 past the barrier; vector-phase slots: 4 top, 5 h1 share written,
6 activation done, 7 h2 stores issued, 8 head partials written, 9 fold done, 10 past the barrier."""
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
st = buf.reshape(256, 8, 16)[:128].astype(np.int64)          # the actor's blocks
for g, nm in ((slice(0, 4), "waves 0-3 (M in phase 16, V in 17)"), (slice(4, 8), "waves 4-7 (V in phase 16, M in 17)")):
    s = st[:, g, :]
    m = (s[..., 1] - s[..., 0]) / 100.0, (s[..., 2] - s[..., 1]) / 100.0, (s[..., 3] - s[..., 2]) / 100.0
    v = [(s[..., b] - s[..., a]) / 100.0 for a, b in ((4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10))]
    print(nm)
    print("   M phase: products %.2f | layer-1 stage + sum %.2f | wait + barrier %.2f us" % tuple(np.median(x) for x in m))
    print("   V phase: h1 share %.2f | bias + activation %.2f | h2 stores %.2f | heads %.2f | fold %.2f | wait + barrier %.2f us" % tuple(np.median(x) for x in v))
agent.close()
