import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cleanrl_jl_amd as crl
med = lambda v: sorted(v)[len(v) // 2]
for nt in (65536, 32768, 131072):
    nbytes = 17 * nt * 128 + 5 * nt
    for tile, seg in ((64, 0), (64, 16), (32, 8), (16, 8)):
        for ntl in (0, 1):
            g, c = crl._lib.gae_bench(nt, 128, seg=seg, tile=tile, nt_loads=ntl, flush_mb=1024, reps=8)
            gm, cm = med(list(g)), med(list(c))
            print(json.dumps({"nts": os.environ.get("CRL_EXP_GAE_NTS"), "nt": nt, "tile": tile, "seg": seg, "nt_loads": ntl, "gae_us": round(gm * 1e3, 1), "copy_us": round(cm * 1e3, 1), "frac": round(nbytes / gm / 1e9 / 8, 3)}), flush=True)
