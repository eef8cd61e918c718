import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cleanrl_jl_amd as crl
med = lambda v: sorted(v)[len(v) // 2]
for nt in (1024, 4096, 8192, 16384, 65536):
    for flush in (0, 1024):
        row = {"nt": nt, "flush": flush}
        for tile, seg in ((0, 0), (128, 0), (128, 16), (256, 0), (256, 16)):
            for ntl in (0, 1):
                g, c = crl._lib.gae_bench(nt, 128, seg=seg, tile=tile, nt_loads=ntl, flush_mb=flush, reps=12)
                row["t%d_s%d_nt%d" % (tile, seg, ntl)] = round(med(list(g)) * 1e3, 1)
        row["copy"] = round(med(list(c)) * 1e3, 1)
        print(json.dumps(row), flush=True)
