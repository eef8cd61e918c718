"""The data-parallel product path with world size > 1 in real, separate processes.
  * RCCL: needs one GPU per rank (the 1-GPU boxes of this project skip it);
  * one-shot peer-mapped all-reduce (csrc/peer.hip): the ranks may share a GPU, so the complete multi-process path — mailbox
    export / IPC attach, [grad | 4 loss sums] exchange per optimiser step, the f64 advantage sums, the guard window's re-run with
    its extra all-reduces — runs on a 1-GPU box too.
Fresh processes are started by torch.distributed.run before anything touches HIP — never a fork / exec of a GPU-initialised
process."""
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


def _run_ranks(world, comm, share_gpu, port, fuse_optim=1):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo", OMP_NUM_THREADS="4", DP2_COMM=comm,
               DP2_SHARE_GPU="1" if share_gpu else "0", DP2_PEER_TIMEOUT_MS="60000", DP2_FUSE_OPTIM=str(fuse_optim))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp2_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DP2_RESULT ")]
    assert line, out.stdout[-2000:]
    r = json.loads(line[0][len("DP2_RESULT "):])
    assert r["replicas_equal_after_steps"]
    assert r["rel_l2_vs_union"] < 1e-5, r            # four Adam steps on the union batch vs on shards + all-reduce
    assert r["iterate"]["replicas_equal"] and r["iterate"]["finite"] and r["iterate"]["exact_reruns"] == 0
    assert r["forced_branch"]["replicas_equal"] and r["forced_branch"]["exact_reruns"] == 3, r
    assert r["iterate"]["stats_equal"] and r["iterate"]["grads_equal"] and r["forced_branch"]["stats_equal"], r
    return r


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_match_union_handle_and_stay_replicated(world):
    if _gpu_count() < world:
        pytest.skip(f"needs {world} GPUs")
    _run_ranks(world, "rccl", False, 29500 + world)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_peer_allreduce_ranks_match_union_handle_and_stay_replicated(world):
    """One GPU per rank when the box has them, otherwise all ranks on GPU 0. Inside crl_ppo_iterate the optimiser step of a peer-attached
    handle is ONE launch (option fuse_optim = 1, the default: reduce -> push -> wait -> rank-order sum -> ClipNorm + Adam,
    update.hip reduce_optim_kernel<true>)."""
    r = _run_ranks(world, "peer", _gpu_count() < world, 29600 + world)
    assert r["iterate"]["fuse_optim"] == 1


def test_one_launch_and_three_launch_data_parallel_steps_agree():
    """The same two-rank run with the optimiser step as ONE launch and as three (reduce_kernel, peer_allreduce_kernel, clipnorm_adam_kernel:
    option fuse_optim = 0): both add the ranks' chunks in rank order, so the all-reduced gradients are the same bits; the per-array norms are
    summed in another grouping, so parameters agree to float32 rounding (the loss records to 1e-6 relative)."""
    share = _gpu_count() < 2
    a = _run_ranks(2, "peer", share, 29621, fuse_optim=1)
    b = _run_ranks(2, "peer", share, 29622, fuse_optim=0)
    assert a["iterate"]["fuse_optim"] == 1 and b["iterate"]["fuse_optim"] == 0
    assert abs(a["iterate"]["loss"] - b["iterate"]["loss"]) <= 1e-6 * abs(b["iterate"]["loss"]) + 5e-7, (a["iterate"], b["iterate"])
    assert abs(a["iterate"]["params_l2"] - b["iterate"]["params_l2"]) <= 1e-6 * b["iterate"]["params_l2"]


def test_bench_runs_multi_rank_on_a_shared_gpu():
    """bench.py's N > 1 path end to end (self-launch, shard, attach, barrier + MAX timing, one JSON line from rank 0)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "peer", "--share-gpu", "--steps", "3",
                          "--warmup", "1", "--total-envs", "4096", "--no-cpu-baseline", "--master-port", "29611"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "shared_gpu" in d["config"]
    # two ranks fit one GPU side by side: the exchange runs inside the one-launch optimiser step (no all-reduce launch of its own to time)
    assert d["config"]["options"]["fuse_optim"] == 1 and d["kernel_ms_per_step"].get("allreduce", 0.0) == 0.0
    assert "peer-mapped" in d["config"]["comm"] and d["last_iteration"]["loss"] == d["last_iteration"]["loss"]      # finite record from the fused step


def test_bench_falls_back_to_the_peer_allreduce_when_rccl_cannot_start():
    """RCCL refuses two ranks on one device; every rank learns that over the rendezvous and all of them attach mailboxes."""
    if _gpu_count() >= 2:
        pytest.skip("needs a box where RCCL cannot place 2 ranks (1 GPU)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "rccl", "--share-gpu", "--steps", "2",
                          "--warmup", "1", "--total-envs", "4096", "--no-cpu-baseline", "--master-port", "29612"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and "RCCL initialisation failed" in d["config"]["comm"]


def test_peer_allreduce_times_out_instead_of_hanging():
    """One rank issues an all-reduce its peer never joins: the kernel gives up after option peer_timeout_ms and the host gets an error."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", DP2_COMM="peer", DP2_SHARE_GPU="1" if _gpu_count() < 2 else "0",
               DP2_MODE="timeout", DP2_PEER_TIMEOUT_MS="2000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", os.path.join(ROOT, "tests", "dp2_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DP2_RESULT ")]
    assert line, out.stdout[-2000:]
    assert "timed out" in json.loads(line[0][len("DP2_RESULT "):])["timeout_error"]
