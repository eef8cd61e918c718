"""The data-parallel product path on REAL RCCL with world size > 1: needs at least two GPUs in the box (the 1-GPU boxes of
this project skip it; tests/test_gpu_dp.py and the forced 1-rank communicator cover the arithmetic there). Fresh processes are
started by torch.distributed.run before anything touches HIP — never a fork / exec of a GPU-initialised process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_count():
    import torch
    return torch.cuda.device_count()     # counting devices does not initialise the GPU on this image


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_match_union_handle_and_stay_replicated(world):
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + world), os.path.join(ROOT, "tests", "dp2_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DP2_RESULT ")]
    assert line, out.stdout[-2000:]
    r = json.loads(line[0][len("DP2_RESULT "):])
    assert r["replicas_equal_after_steps"]
    assert r["rel_l2_vs_union"] < 1e-5, r            # four Adam steps on the union batch vs on shards + all-reduce
    assert r["iterate"]["replicas_equal"] and r["iterate"]["finite"] and r["iterate"]["exact_reruns"] == 0
    assert r["forced_branch"]["replicas_equal"] and r["forced_branch"]["exact_reruns"] == 3, r
