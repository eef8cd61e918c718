"""Timeline of the producer / consumer forward kernel (ninth tile of every block; build: bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402
L = crl._lib
agent = crl.Agent(crl.PPOConfig(num_envs=16384, num_steps=128, total_timesteps=16384 * 128 * 100), obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC,
                  options={"wide_fuse": 1})   # forward only: the backward kernel shares the stamp slots
h = agent.handle; h.env_reset(); h.iterate(1, want_stats=False); h.sync()
lib = L.load(); buf = np.zeros(2 * 256 * 8 * 16, np.uint64)
lib.crl_debug_read_wstamps.argtypes = [C.c_void_p, C.c_int32]
assert lib.crl_debug_read_wstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(2, 256, 8, 16)[1, :128, :, :16].astype(np.int64)
ok = (st[:, :, 0] > 0).all(axis=1); st = st[ok]
us = (st - st[:, :, :1].min(axis=1, keepdims=True)) / 100.0
print(len(st), "blocks; us since B_start of the ninth tile")
for g, nm in ((slice(0, 4), "consumers"), (slice(4, 8), "producers")):
    u = us[:, g, :]
    print("  %-10s B_start %5.2f | top of slab 3 %5.2f | own slab-3 work done %5.2f | loop end %5.2f | before B_epi %5.2f | after B_epi %5.2f" %
          (nm, np.median(u[..., 0]), np.median(u[..., 11]), np.median(u[..., 4]), np.median(u[..., 1]), np.median(u[..., 2]), np.median(u[..., 3])))
agent.close()
p = us[:, 4:8, :]
c = us[:, 0:4, :]
print("  slab 3, producers: top %.2f | bias loads / DMA issue +%.2f | tanh, split, LDS stores issued +%.2f | waits done +%.2f | next top +%.2f" % (
    np.median(p[..., 11]), np.median(p[..., 6] - p[..., 11]), np.median(p[..., 4] - p[..., 6]), np.median(p[..., 7] - p[..., 4]), np.median(p[..., 12] - p[..., 7])))
print("  slab 3, consumers: top %.2f | 48 MFMAs (+ 8 DMA pieces) issued +%.2f | vmcnt wait done +%.2f | next top +%.2f" % (
    np.median(c[..., 11]), np.median(c[..., 4] - c[..., 11]), np.median(c[..., 7] - c[..., 4]), np.median(c[..., 12] - c[..., 7])))
print("  top of slab 0..7 and loop end:", " ".join("%.2f" % np.median(p[..., 8 + s]) for s in range(8)), "%.2f" % np.median(p[..., 1]))
raw = buf.reshape(2, 256, 8, 16)[1, :128][ok].astype(np.int64)[:, :4, :]
print("  shader clock during the slab loop (s_memtime / s_memrealtime, consumer waves): %.0f MHz" % np.median((raw[..., 6] - raw[..., 5]) / ((raw[..., 1] - raw[..., 0]) / 100.0)))
