#!/usr/bin/env python3
"""How much instruction-level parallelism the update kernel's tile loops offer a wave: for every vector instruction of a role's loop
(-DCRL_COUNT_PROBE build, as scripts/count_isa.py), the distance in issued instructions to the nearest earlier instruction that wrote
one of its source registers. scripts/micro/inst_rate.hip measures what that distance costs on gfx950: a lone wave issues an
independent v_fma_f32 every 2.4 ns but a back-to-back dependent one every 4.0 ns (two waves per SIMD: 1.33 vs 2.0 ns).
    python scripts/dep_distance.py            → histogram per role + the longest back-to-back dependent runs"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cleanrl.jl_amd", "csrc")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def main():
    extra = sys.argv[1:]
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only",
                          "-DCRL_COUNT_PROBE", *extra, "-S", os.path.join(CSRC, "update.hip"), "-o", "-"], capture_output=True, text=True, check=True).stdout.split("\n")
    start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN3crl16update_x2_kernel\S*:", l))
    end = next(i for i in range(start, len(asm)) if ".amdhsa_kernel" in asm[i] or asm[i].startswith(".Lfunc_end"))
    labels = {m.group(1): i for i in range(start, end) for m in [re.match(r"^(\.LBB\d+_\d+):", asm[i])] if m}
    loops = []
    for i in range(start, end):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", asm[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    n = lambda a, b, pat: sum(1 for i in range(a, b + 1) if pat in asm[i])
    by_header = {}
    for a, b in loops:
        if n(a, b, "v_mfma_f32_32x32x16_f16") == 72:
            by_header[a] = max(by_header.get(a, a), b)
    tl = sorted(by_header.items())
    tl = [(a, b) for a, b in tl if not any((x <= a and b <= y) and (x, y) != (a, b) for x, y in tl)]
    for a, b in tl:
        role = "actor" if n(a, b, "v_log_f32") else "critic"
        last = {}                       # register -> index of the instruction that wrote it last
        hist = Counter(); idx = 0; runs = []; run = 0; run_start = 0
        by_op = Counter()
        for i in range(a, b + 1):
            l = asm[i].split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            if not op.startswith(("v_", "ds_", "global_", "buffer_")):
                continue                # scalar instructions issue from another port
            idx += 1
            toks = l[len(op):].split(",")
            nd = 1
            dst = regs(toks[0]) if toks else []
            if op.startswith(("ds_write", "ds_store", "global_store", "global_load_lds", "buffer_store")) or op.startswith("v_cmp"):
                dst = []; nd = 0
            src = [r for t in toks[nd:] for r in regs(t)]
            if op.startswith(("v_mfma", "v_fmac", "v_mac", "v_pk_fmac")) or "fmac" in op:
                src += dst
            if op.startswith("v_") and not op.startswith("v_mfma"):
                d = min((idx - last[r] for r in src if r in last), default=99)
                hist[min(d, 8)] += 1
                if d == 1:
                    by_op[op] += 1
                    if run == 0:
                        run_start = i
                    run += 1
                else:
                    if run >= 6:
                        runs.append((run, run_start - start))
                    run = 0
            for r in dst:
                last[r] = idx
        tot = sum(hist.values())
        print(f"{role}: {tot} vector instructions; distance to the producer of a source: " +
              ", ".join(f"{k if k < 8 else '8+'}: {100.0 * v / tot:.1f}%" for k, v in sorted(hist.items())))
        print("   back-to-back dependent (distance 1) by opcode:", dict(by_op.most_common(12)))
        runs.sort(reverse=True)
        print("   longest dependent runs (length, line offset in the kernel):", runs[:10])


if __name__ == "__main__":
    main()
