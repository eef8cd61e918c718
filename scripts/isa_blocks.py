#!/usr/bin/env python3
"""Per-basic-block instruction histogram of one kernel of csrc/<file>.hip (cross-compiled ISA): python scripts/isa_blocks.py wide.hip <mangled-name substring> [min ops]"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "cleanrl.jl_amd", "csrc", sys.argv[1])
out = "/tmp/isa_%s.s" % os.path.basename(src)
if not (len(sys.argv) > 4 and sys.argv[4] == "reuse"):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only", "-S", src, "-o", out] + os.environ.get("EXTRA", "").split(), capture_output=True)
s = open(out).read()
key = sys.argv[2]
minops = int(sys.argv[3]) if len(sys.argv) > 3 else 20
starts = [i for i in range(len(s)) if s.startswith("\n_Z", i) and key in s[i + 1:s.index(":", i)]]
for st in starts:
    name = s[st + 1:s.index(":", st)]
    e = s.index(".Lfunc_end", st)
    print("==", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:120])
    blk = "entry"; cnt = collections.OrderedDict(); cnt[blk] = collections.Counter()
    for l in s[st:e].splitlines():
        ls = l.strip()
        if ls.startswith(".LBB"):
            blk = ls.split(":")[0] + (" (loop)" if "Loop Header" in ls else ""); cnt[blk] = collections.Counter(); continue
        if ls and not ls.startswith(";") and not ls.startswith(".") and not ls.endswith(":"):
            op = ls.split()[0]
            k = ("mfma" if "mfma" in op else "ds_read" if op.startswith("ds_read") else "ds_write" if op.startswith("ds_write") else "scratch" if op.startswith("scratch")
                 else "global" if op.startswith("global") else "barrier" if "barrier" in op else "trans" if op.split("_")[1] in ("exp", "rcp", "log", "rsq", "sqrt") and op.startswith("v_") else "valu" if op.startswith("v_") else "waitcnt" if op == "s_waitcnt" else "salu" if op.startswith("s_") else op)
            cnt[blk][k] += 1
    for b, c in cnt.items():
        if sum(c.values()) >= minops: print("  ", b, dict(c))
