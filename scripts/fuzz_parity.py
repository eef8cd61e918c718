#!/usr/bin/env python3
"""Randomised whole-iteration parity from a shell (the case itself: tests/fuzzlib.py; 40 cases of seed 1 run in the GPU test suite as
tests/test_gpu_fuzz.py). Prints one line per configuration and a summary; exit code 1 on any mismatch.
Usage: python scripts/fuzz_parity.py [n_configs] [seed]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cleanrl_jl_amd as crl          # noqa: E402
import fuzzlib                         # noqa: E402
import oraclelib as O                  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = []
for case in range(N):
    line = fuzzlib.run_case(crl, O, rng, case)
    print(json.dumps(line), flush=True)
    if not line["ok"]:
        fails.append(line)
print(json.dumps({"configs": N, "failed": len(fails)}))
sys.exit(1 if fails else 0)
