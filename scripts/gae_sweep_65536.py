"""Cold sweep of the standalone GAE kernels at the size BASELINE's metric is quoted on (65536 envs x 128 steps, 143 MB) beside the library's own
non-temporal copy of the same bytes (crl_gae_bench): segmented kernel (tile = envs per block 16/32/64, seg = steps per segment 8/16), its two-envs-per-thread form (tile 128 / 256 = 32 / 64 env pairs per block) and streaming kernel
(tile 4/2/1 = envs per thread, seg = window depth).   python scripts/gae_sweep_65536.py"""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cleanrl_jl_amd as crl
med = lambda v: sorted(v)[len(v) // 2]
for nt in (65536, 32768, 131072, 262144):
    nbytes = 17 * nt * 128 + 5 * nt
    for tile, seg in ((0, 0), (64, 16), (32, 8), (128, 8), (128, 16), (256, 8), (256, 16), (2, 8), (1, 16)):
        for ntl in (0, 1):
            try:
                g, c = crl._lib.gae_bench(nt, 128, seg=seg, tile=tile, nt_loads=ntl, flush_mb=1024, reps=10)
            except Exception as e:   # noqa: BLE001
                print(json.dumps({"nt": nt, "tile": tile, "seg": seg, "error": str(e)[:80]})); continue
            gm, cm = med(list(g)), med(list(c))
            print(json.dumps({"nt": nt, "tile": tile, "seg": seg, "nt_loads": ntl, "gae_us": round(gm * 1e3, 1), "copy_us": round(cm * 1e3, 1), "frac": round(nbytes / gm / 1e9 / 8, 3), "copy_TBs": round(nbytes / cm / 1e9, 2)}), flush=True)
