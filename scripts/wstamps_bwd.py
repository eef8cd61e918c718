"""Timeline of the fused backward kernel's ninth tile (builds: bash scripts/build_variant.sh wstamps "-DCRL_EXP_WSTAMPS -DCRL_BTS_SET=0" wide ; wstamps1 with SET=1).
CRL_LIB_PATH=cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so python scripts/wstamps_bwd.py <set>"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402
L = crl._lib
SET = int(sys.argv[1]) if len(sys.argv) > 1 else 0
agent = crl.Agent(crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100), obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC)
h = agent.handle; h.env_reset(); h.iterate(1, want_stats=False); h.sync()
lib = L.load(); buf = np.zeros(2 * 256 * 8 * 16, np.uint64)
lib.crl_debug_read_wstamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_wstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = (buf.reshape(2, 256, 8, 16)[1, :128] & 0xFFFFFFFF).astype(np.int64)
order = ([0, 10, 1], [2, 7, 8, 9, 3, 4], [5, 11, 6])[SET]
names = ({0: "tile top", 10: "first transfers landed + barrier", 1: "slab 0 staged + barrier (loop starts)"},
         {2: "top of slab 3", 7: "weight DMA of slab 4 issued", 8: "first phase done (sg 0: staging, sg 1: MFMAs)", 9: "second marker (h2 DMA issued)", 3: "own work of slab 3 done", 4: "top of slab 4 (after wait + barrier)"},
         {5: "slab loop done", 11: "epilogue set-up + barrier", 6: "tile end (epilogue done + barrier)"})[SET]
ok = (st[:, :, order[0]] > 0).all(axis=1); st = st[ok]
t0 = st[:, :, order[0]].min(axis=1, keepdims=True)
print(f"{len(st)} blocks, ninth tile; us since the block's first wave reached '{names[order[0]]}' (median over blocks; sg 0 = waves 0-3 stage then multiply, sg 1 = waves 4-7 multiply then stage)")
for g, nm in ((slice(0, 4), "sg 0"), (slice(4, 8), "sg 1")):
    for k in order:
        u = (st[:, g, k] - t0) / 100.0
        print("   %s  %-50s %6.2f" % (nm, names[k], np.median(u)))
agent.close()
