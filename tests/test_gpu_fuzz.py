"""Randomised whole-iteration parity in the GPU suite: 40 random small configurations (seed 1) of tests/fuzzlib.py — permutations and
actions bit-equal, losses within 2e-6 relative, parameters within 1e-5 relative L2 — against the CPU oracle."""
import numpy as np
import pytest

import fuzzlib
import oraclelib as O

pytestmark = pytest.mark.gpu


def test_forty_random_configurations_match_the_oracle():
    import cleanrl_jl_amd as crl
    rng = np.random.default_rng(1)
    lines = [fuzzlib.run_case(crl, O, rng, case) for case in range(40)]
    bad = [l for l in lines if not l["ok"]]
    assert not bad, bad[:3]
    assert max(l["loss_rel"] for l in lines) <= fuzzlib.LOSS
    # the sweep must have covered both drivers and both GAE modes
    assert {l["blocked"] for l in lines} == {True, False} and {l["gae_mode"] for l in lines} == {0, 1}
