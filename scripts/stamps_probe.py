"""In-kernel timeline of update_x2_kernel (diagnostic build: bash scripts/build_variant.sh stamps -DCRL_EXP_STAMPS "update").
CRL_LIB_PATH=cleanrl.jl_amd/variants/stamps/libcleanrl_hip.so python scripts/stamps_probe.py [num_envs]
Stamps per wave (100 MHz wall clock): 0 entry, 1 weights staged, 2 tile loop done, 3 block reduction done."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=128, total_timesteps=nt * 128 * 100))
h = agent.handle
h.env_reset()
h.iterate(3, want_stats=False)
h.sync()
L = crl._lib.load()
buf = np.zeros(512 * 8 * 8, np.uint64)
L.crl_debug_read_stamps.argtypes = [C.c_void_p, C.c_int32]
assert L.crl_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(512, 8, 8)[:, :, :4].astype(np.int64)
live = st[:, :, 0] > 0
nb = int(live[:, 0].sum())
st = st[:nb]
t0 = st[:, :, 0].min()
us = (st - t0) / 100.0
print(f"num_envs {nt}: {nb} blocks of the LAST update launch; times in us from the first wave's entry")
for name, sl in (("older waves 0-3", slice(0, 4)), ("younger waves 4-7", slice(4, 8))):
    u = us[:, sl, :]
    print(f"  {name}: entry {np.median(u[..., 0]):6.2f} (max {u[..., 0].max():6.2f}) | staged {np.median(u[..., 1]):6.2f} | tiles done {np.median(u[..., 2]):6.2f} "
          f"(min {u[..., 2].min():6.2f} max {u[..., 2].max():6.2f}) | reduced {np.median(u[..., 3]):6.2f} (max {u[..., 3].max():6.2f})")
print(f"  stage {np.median(us[..., 1] - us[..., 0]):.2f} us, tile loop {np.median(us[..., 2] - us[..., 1]):.2f} us (older {np.median(us[:, :4, 2] - us[:, :4, 1]):.2f}, "
      f"younger {np.median(us[:, 4:, 2] - us[:, 4:, 1]):.2f}), block end after last wave's loop {np.median(us[..., 3].max(axis=1) - us[..., 2].max(axis=1)):.2f} us, "
      f"launch span {us[..., 3].max():.2f} us")
agent.close()
