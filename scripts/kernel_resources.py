#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in one csrc/*.hip file, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950 without a GPU).  python scripts/kernel_resources.py update.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "cleanrl.jl_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + os.environ.get("EXTRA", "").split()
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void crl::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(\w[\w ]*\w|\w+)\s*(?:\[bytes/lane\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    if flt in k:
        print(f"{k:60s} VGPR {v.get('VGPRs','?'):>4} AGPR {v.get('AGPRs','?'):>4} SGPR {v.get('TotalSGPRs', v.get('SGPRs','?')):>4} "
              f"scratch {v.get('ScratchSize','?'):>5} occ {v.get('Occupancy','?'):>2} LDS {v.get('LDS Size','?')}")
