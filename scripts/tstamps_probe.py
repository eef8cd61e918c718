"""Phase timeline of ONE tile (each wave's 21st) of update_x2_kernel (diagnostic build: bash scripts/build_variant.sh stamps -DCRL_EXP_STAMPS update).
CRL_LIB_PATH=cleanrl.jl_amd/variants/stamps/libcleanrl_hip.so python scripts/tstamps_probe.py [num_envs]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=128, total_timesteps=nt * 128 * 100))
h = agent.handle; h.env_reset(); h.iterate(3, want_stats=False); h.sync()
L = crl._lib.load(); buf = np.zeros(512 * 8 * 16, np.uint32)
L.crl_debug_read_tstamps.argtypes = [C.c_void_p, C.c_int32]
assert L.crl_debug_read_tstamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(512, 8, 16).astype(np.int64)
names = ["tile top -> records in registers (vmcnt wait, LDS reads, next DMA issued)", "forward: layer 1, act, layer 2 (24 MFMAs), act, head", "loss terms, output cotangent",
         "(1)+(2) h2T to LDS, dW3 / db3 sums", "(3) delta2 = W3T d3 * (1 - h2^2)", "(4) sample scale, split, W2T delta2 (24 MFMAs), delta1", "(7) delta1T to LDS, db1 / dW1 sums",
         "(5) delta2T to LDS, db2, B fragments", "(6) h1T to LDS, splits, dW2 (24 MFMAs)"]
for role, rn in ((1, "actor"), (2, "critic")):
    sel = st[..., 15] == role
    s = st[sel][:, :10]
    if not len(s):
        continue
    have = [k for k in range(10) if (s[:, k] != 0).any()]          # the build's stamp set
    tot = (s[:, 9] - s[:, 0]) / 100.0
    print(f"{rn}: {len(s)} waves, tile {np.median(tot):.2f} us (median; p10 {np.percentile(tot, 10):.2f}, p90 {np.percentile(tot, 90):.2f}); stamped points {have}")
    for a, b in zip(have[:-1], have[1:]):
        d = (s[:, b] - s[:, a]) / 100.0
        print("   %-90s %5.2f us  (p10 %.2f p90 %.2f)  %4.1f %%" % (" + ".join(names[a:b]) if b - a <= 2 else f"{names[a]} ... {names[b - 1]}", np.median(d), np.percentile(d, 10), np.percentile(d, 90),
                                                                      100 * np.median(d) / np.median(tot)))
agent.close()
