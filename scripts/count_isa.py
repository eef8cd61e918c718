#!/usr/bin/env python3
"""Static instruction counts of the update kernel's tile loop, from the gfx950 ISA hipcc emits (cross-compiles without a GPU).

update.hip is compiled with -DCRL_COUNT_PROBE, which leaves out the cold paths (the in-loop bf16x3 weight-gradient fallback of a tile
whose cotangents fall outside the launch's fp16 window, and the per-role bf16x3 fallback for |w| >= 255 — wave-uniform branches the
default workload does not take), so each role's tile loop of update_x2_kernel is one plain loop holding the 72 f16 MFMAs of a tile
(24 forward + 24 backward-data + 24 weight-gradient); the role with v_log_f32 (the softmax) is the actor. Instructions are
classed as VALU (transcendental ones separately: v_exp/v_rcp/v_log/v_rsq/v_sqrt issue at a quarter of the plain rate, Float64
and packed ones at half), MFMA, LDS, VMEM, SALU/other. Writes profiles/<tag>_update_kernel_isa.json with a hash of the kernel sources.
bench.py turns `issue_slots_per_tile` × tiles ÷ launch time into the valu-issue roofline; the PMC pass (SQ_INSTS_VALU, final_measure.sh)
is the cross-check.     python scripts/count_isa.py <tag>"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cleanrl.jl_amd", "csrc")


def source_hash():
    """sha256 over the kernel sources: a profile taken with other sources is stale."""
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


TRANS = re.compile(r"^v_(exp|rcp|log|rsq|sqrt|sin|cos)_")
HALF_RATE = re.compile(r"^v_(pk_(fma|mul|add)_f32|(fma|mul|add|min|max|rndne|fract|trunc|ceil|floor|ldexp|frexp_mant|div_fmas|div_fixup|div_scale)_f64|cvt_f64_|cvt_f32_f64|cvt_i32_f64|cvt_u32_f64|cmp_\w+_f64)")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"      # a spill inside the tile loop: should not exist
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        if TRANS.match(op):
            return "valu_trans"
        if HALF_RATE.match(op):
            return "valu_half"
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only", "-DCRL_COUNT_PROBE",
                          "-S", os.path.join(CSRC, "update.hip"), "-o", "-"], capture_output=True, text=True, check=True).stdout.split("\n")
    start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN3crl16update_x2_kernel\S*:", l))
    end = next(i for i in range(start, len(asm)) if ".amdhsa_kernel" in asm[i] or asm[i].startswith(".Lfunc_end"))
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", asm[i])
        if m:
            labels[m.group(1)] = i
    loops = []
    for i in range(start, end):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", asm[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))

    def ops(a, b):
        out = []
        for i in range(a, b + 1):
            l = asm[i].strip()
            if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
                continue
            out.append((i, l.split()[0]))
        return out

    def has(a, b, pat):
        return any(pat in asm[i] for i in range(a, b + 1))

    def n(a, b, pat):
        return sum(1 for i in range(a, b + 1) if pat in asm[i])

    # the tile loop of a role: every f16 MFMA of a tile (24 forward + 24 backward-data + 24 weight-gradient). Several back-edges
    # may share the header: the longest range is the loop.
    by_header = {}
    for a, b in loops:
        if n(a, b, "v_mfma_f32_32x32x16_f16") == 72 and n(a, b, "v_mfma_f32_32x32x16_bf16") == 0:
            by_header[a] = max(by_header.get(a, a), b)
    tile_loops = sorted(by_header.items())
    tile_loops = [(a, b) for a, b in tile_loops if not any((x <= a and b <= y) and (x, y) != (a, b) for x, y in tile_loops)]   # outermost only
    if len(tile_loops) != 2:
        print(f"count_isa: expected the two tile loops of update_x2_kernel (actor, critic), found {len(tile_loops)}: the code layout changed", file=sys.stderr)
    roles = {}
    for a, b in tile_loops:
        name = "actor" if n(a, b, "v_log_f32") else "critic"
        skip = (0, -1); bf = []
        cnt = {}
        for i, op in ops(a, b):
            if skip[0] <= i <= skip[1]:
                continue
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
        valu_all = cnt.get("valu", 0) + cnt.get("valu_trans", 0) + cnt.get("valu_half", 0)
        # issue slots in units of one plain wave64 VALU instruction (2 cycles on a SIMD-32, MI355X_MICROARCH.md "v_fma_f32 (wave64)
        # 2 cyc"): transcendental 4x the plain issue cost in that table (8 vs 4 for a lone wave) → 2 slots is the conservative reading
        # used here; half-rate Float64 / packed 2; an MFMA holds the vector issue port for 8 of its 32 cycles → 2 slots (8 of a lone
        # wave's 4-cycle slots = 2)
        slots = cnt.get("valu", 0) + 2 * cnt.get("valu_trans", 0) + 2 * cnt.get("valu_half", 0) + 2 * cnt.get("mfma", 0)
        roles[name] = {"lines": [a - start, b - start], "fallback_region_lines": [skip[0] - start, skip[1] - start] if bf else None,
                       "counts": cnt, "valu_total": valu_all, "issue_slots_per_tile": slots}
    out = {"kernel": "update_x2_kernel<4, 2>", "source_hash": source_hash(), "roles": roles,
           "note": "static count over the tile loop of each role (the in-loop bf16x3 fallback lies outside the loop's range); exec-masked regions are counted "
                   "(they issue); a slot = one plain wave64 VALU issue (2 cycles of a SIMD-32). CAVEAT: a static count covers every block inside the loop's "
                   "address range — where the compiler has cloned part of the body (loop unswitching on a wave-uniform condition) it over-counts; "
                   "the PMC counts of scripts/final_measure.sh (SQ_INSTS_VALU per launch) are what bench.py uses"}
    path = os.path.join(ROOT, "profiles", f"{tag}_update_kernel_isa.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
