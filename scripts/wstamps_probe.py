"""In-kernel timeline of the fused 2x256 kernels (diagnostic build: bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide).
CRL_LIB_PATH=cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so python scripts/wstamps_probe.py
slots: 0 tile entry, 1 first slab staged + barrier, 2 top of slab 3, 3 own work of slab 3 done, 4 top of slab 4 (after wait + barrier), 5 loop done, 6 epilogue done"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402

L = crl._lib
cfg = crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100)
agent = crl.Agent(cfg, obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
h = agent.handle
h.env_reset(); h.iterate(1, want_stats=False); h.sync()
lib = L.load()
buf = np.zeros(2 * 256 * 8 * 16, np.uint64)
lib.crl_debug_read_wstamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_wstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(2, 256, 8, 16)[..., :12].astype(np.int64)
for k, name in enumerate(("forward", "backward")):
    s = st[k]
    ok = (s[:, :, 0] > 0).all(axis=1)
    s = s[ok]
    if len(s) == 0:
        continue   # (the symmetric forward kernel is not the default any more: no stamps of it)
    us = (s - s[:, :, :1].min(axis=1, keepdims=True)) / 100.0     # per block: from its first wave's entry
    print(f"{name}: {len(s)} blocks (first tile of each); median over blocks, us since the block's first wave entered the tile")
    for g, nm in ((slice(0, 4), "waves 0-3 (sg 0: stage then compute)"), (slice(4, 8), "waves 4-7 (sg 1: compute then stage)")):
        u = us[:, g, :]
        print("   %-40s" % nm, " ".join("%s %6.2f" % (lab, np.median(u[..., i])) for i, lab in enumerate(("entry", "staged0", "top3", "own3", "top4", "loopend", "epiend"))))
    for g, nm in ((slice(0, 4), "sg0"), (slice(4, 8), "sg1")):
        u = us[:, g, :]
        print(f"   slab 3, {nm}: top → weight DMA issued {np.median(u[..., 7] - u[..., 2]):.2f} | → first phase done {np.median(u[..., 8] - u[..., 7]):.2f} | "
              f"→ (bwd) second marker {np.median(u[..., 9] - u[..., 8]):.2f} | → own work done {np.median(u[..., 3] - np.maximum(u[..., 8], u[..., 9])):.2f}")
    d = np.median(us[..., 5] - us[..., 1]); print(f"   slab loop {d:.2f} us ({d / 8:.2f} per slab); slab 3 own work {np.median(us[..., 3] - us[..., 2]):.2f}, wait+barrier {np.median(us[..., 4] - us[..., 3]):.2f}; "
                                                  f"prologue {np.median(us[..., 1]):.2f}, epilogue {np.median(us[..., 6] - us[..., 5]):.2f}")
s = st[1]; s = s[(s[:, :, 0] > 0).all(axis=1)]
us = (s - s[:, :, :1].min(axis=1, keepdims=True)) / 100.0
print("backward tile (ninth of the block; LDS-resident stamps): top %.2f | first transfers landed + barrier %.2f | slab 0 staged + barrier %.2f | loop end %.2f | "
      "epilogue set-up + barrier %.2f | tile end %.2f" % tuple(np.median(us[..., i]) for i in (0, 10, 1, 5, 11, 6)))
agent.close()
