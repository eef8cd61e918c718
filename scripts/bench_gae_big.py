"""Standalone GAE kernel on inputs larger than the Infinity Cache (crl_gae_bench): sweep of tile / segment / load flavour at
nt = 262144 and 524288 (0.57 / 1.14 GB per launch), each next to the hand-written float4 copy of the same byte count.
python scripts/bench_gae_big.py [nt,nt,…]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cleanrl_jl_amd as crl   # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [65536, 262144, 524288]
med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
for nt in sizes:
    nbytes = 17 * nt * 128 + 5 * nt
    for tile, seg in ((4, 4), (4, 8), (4, 16), (2, 4), (2, 8), (2, 16), (1, 8), (1, 16), (64, 0), (64, 16), (0, 0)):
        if True:
            for ntl in (0, 1):
                try:
                    g, c = crl._lib.gae_bench(nt, 128, seg=seg, tile=tile, nt_loads=ntl, flush_mb=(1024 if nbytes < (300 << 20) else 0), reps=8)
                except crl._lib.CrlError as e:
                    print(json.dumps({"nt": nt, "tile": tile, "seg": seg, "nt_loads": ntl, "error": str(e)})); continue
                gm, cm = med(list(g)), med(list(c))
                print(json.dumps({"nt": nt, "tile": tile, "seg": seg, "nt_loads": ntl, "gae_us": round(gm * 1e3, 1), "copy_us": round(cm * 1e3, 1),
                                  "gae_TBps": round(nbytes / gm / 1e9, 3), "frac_of_8TBps": round(nbytes / gm / 1e9 / 8, 3), "over_copy": round(cm / gm, 3)}), flush=True)
