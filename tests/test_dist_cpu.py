"""N>1 path on CPU: world_size-2 gloo processes check the sharding contract the multi-GPU run relies on, with the CPU
oracle as the checker (the HIP path itself needs GPUs):
  * shard r's rollout == envs [r*n,(r+1)*n) of the single-process rollout (global env ids key the RNG streams);
  * Σ_ranks of gradients computed with the GLOBAL minibatch size and GLOBAL advantage statistics == the single-process
    gradient of the union minibatch (the all-reduce the library performs, ppo.jl:250 cadence);
  * the 128-byte communicator id reaches every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oraclelib as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total_envs, k, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    distmod = importlib.import_module("cleanrl_jl_amd.dist")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = distmod.exchange_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        n, off = distmod.shard_envs(total_envs, world, rank)
        cfg = O.make_config(num_envs=n, num_steps=k, env_id_offset=off)
        st = O.State(cfg)
        full_cfg = O.make_config(num_envs=total_envs, num_steps=k)
        params = O.orthogonal_params(full_cfg, 0)
        st.params[:] = params
        st.env_init(); st.rollout(); st.compute_gae()
        # --- shard invariance of the rollout
        parts = [torch.zeros(n * k, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(st.action.ravel(order="F"))))
        obs_parts = [torch.zeros(4 * n * k) for _ in range(world)]
        dist.all_gather(obs_parts, torch.from_numpy(np.ascontiguousarray(st.obs.ravel(order="F"))))
        # --- data-parallel gradient: local minibatch = first quarter of the local batch
        B = n * k; M = B // 4
        mb = np.arange(M, dtype=np.int32)
        sums = torch.tensor([float(st.adv.ravel(order="F")[mb].astype(np.float64).sum()),
                             float((st.adv.ravel(order="F")[mb].astype(np.float64) ** 2).sum())], dtype=torch.float64)
        dist.all_reduce(sums)
        Mg = distmod.global_minibatch(M, world)
        mean = sums[0].item() / Mg
        std = np.sqrt(max((sums[1].item() - Mg * mean * mean) / (Mg - 1), 0.0))
        g, s = O.loss_grad(cfg, params, st.obs.reshape(4, -1, order="F"), st.action, st.logprob, st.value, st.adv, st.ret, mb,
                           adv_stats=[np.float32(mean), np.float32(std)])
        gt = torch.from_numpy(g.astype(np.float64) / world)   # local means → global mean
        dist.all_reduce(gt)
        if rank == 0:
            fst = O.State(full_cfg); fst.params[:] = params
            fst.env_init(); fst.rollout(); fst.compute_gae()
            act = fst.action
            for r in range(world):
                assert np.array_equal(parts[r].numpy().reshape(n, k, order="F"), act[r * n:(r + 1) * n]), "shard rollout differs"
                assert np.array_equal(obs_parts[r].numpy().reshape(4, n, k, order="F"), fst.obs[:, r * n:(r + 1) * n])
            # union minibatch in the full batch's flat indexing b = e + nt*t
            e = np.arange(B) % n; t = np.arange(B) // n
            union = np.concatenate([(r * n + e[:M]) + total_envs * t[:M] for r in range(world)]).astype(np.int32)
            gf, sf = O.loss_grad(full_cfg, params, fst.obs.reshape(4, -1, order="F"), fst.action, fst.logprob, fst.value,
                                 fst.adv, fst.ret, union)
            assert abs(sf["adv_mean"] - np.float32(mean)) < 1e-6 and abs(sf["adv_std"] - np.float32(std)) < 1e-5
            err = np.linalg.norm(gt.numpy() - gf) / np.linalg.norm(gf)
            assert err < 1e-5, err
            open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    O.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 16, 32, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_envs_contract():
    import importlib
    distmod = importlib.import_module("cleanrl_jl_amd.dist")
    assert distmod.shard_envs(65536, 8, 3) == (8192, 24576)
    assert distmod.global_minibatch(262144, 8) == 2097152
    with pytest.raises(ValueError):
        distmod.shard_envs(10, 4, 0)
    with pytest.raises(ValueError):
        distmod.shard_envs(8, 2, 2)


def _bench(*args, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, env=env,
                          timeout=timeout)


def test_bench_dry_run_prints_one_fresh_process_per_gpu():
    """`python bench.py --gpus N` as a bare command: the parent (which never touches the GPU) starts the launcher as a CHILD;
    --dry-run shows that command and every rank's environment."""
    import json
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-run")
    assert out.returncode == 0, out.stderr
    lines = [json.loads(l) for l in out.stdout.strip().splitlines()]
    launcher = lines[0]["launcher"]
    assert launcher[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in launcher and "127.0.0.1" in launcher
    assert launcher[-6:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"] and "--dry-run" not in launcher
    ranks = lines[1:]
    assert [r["rank"] for r in ranks] == [0, 1]
    for r in ranks:
        assert r["env"]["WORLD_SIZE"] == "2" and r["env"]["LOCAL_RANK"] == str(r["rank"]) and r["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert r["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert r["cmd"][-6:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"]


def test_bench_bare_command_spawns_ranks_and_exchanges_the_id():
    """The real spawn path on CPU: bare `bench.py --gpus 2 --rendezvous-only` → child launcher → two ranks meet over gloo,
    rank 0's 128-byte communicator id reaches rank 1, the env axis is sharded, rank 0 prints ONE JSON line."""
    import json
    out = _bench("--gpus", "2", "--rendezvous-only", "--total-envs", "4096")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"rendezvous": "ok", "world": 2, "envs_per_rank": 2048}


class _FakePeerHandle:
    """Records what the launcher-side plumbing of the one-shot peer all-reduce hands to the C ABI."""
    def __init__(self, rank):
        self.rank = rank; self.attached = None

    def comm_peer_export(self, world, rank):
        return bytes([rank + 1]) * 64

    def comm_peer_attach(self, handles):
        self.attached = handles


def _peer_worker(rank, world, port, out_dir):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    crl_dist = importlib.import_module("cleanrl_jl_amd.dist")
    h = _FakePeerHandle(rank)
    crl_dist.attach_comm(dist, h, world, rank, "peer", None)
    np.save(os.path.join(out_dir, f"peer{rank}.npy"), np.frombuffer(h.attached, dtype=np.uint8))
    try:
        crl_dist.attach_comm(dist, h, world, rank, "ring", None)
        ok = False
    except ValueError:
        ok = True
    assert ok
    dist.destroy_process_group()


def test_peer_mailbox_handles_are_gathered_in_rank_order(tmp_path):
    """crl_comm_peer_attach wants world_size x 64 bytes in rank order on every rank (include/cleanrl_hip.h)."""
    world = 3
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = np.concatenate([np.full(64, r + 1, np.uint8) for r in range(world)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"peer{r}.npy"), want)


class _FakeCommHandle(_FakePeerHandle):
    """A handle whose RCCL start can be made to fail per rank; like the real one, peer_export refuses while a communicator is up."""
    def __init__(self, rank, rccl_fails):
        super().__init__(rank)
        self.rccl_fails = rccl_fails; self.comm_up = False; self.destroyed = 0

    def comm_init(self, uid, world, rank):
        if self.rccl_fails:
            raise RuntimeError("ncclCommInitRank failed")
        self.comm_up = True

    def comm_destroy(self):
        self.comm_up = False; self.destroyed += 1

    def comm_peer_export(self, world, rank):
        if self.comm_up:
            raise RuntimeError("a communicator is already attached")
        return super().comm_peer_export(world, rank)


def _fallback_worker(rank, world, port, out_dir, case):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    crl_dist = importlib.import_module("cleanrl_jl_amd.dist")
    if case == "no_id":          # librccl cannot even be loaded on rank 0: make_id raises there, nobody else may hang
        def make_id():
            raise RuntimeError("cannot load librccl")
        h = _FakeCommHandle(rank, rccl_fails=False)
    elif case == "no_lib_rank1":  # librccl loads on rank 0 only: rank 0 must NOT enter comm_init (a collective that would never return)
        def make_id():
            if rank == 1:
                raise RuntimeError("cannot load librccl")
            return bytes(range(128))

        class _NeverInit(_FakeCommHandle):
            def comm_init(self, uid, world, rank):
                raise AssertionError("comm_init entered although a rank cannot load librccl: it would block in ncclCommInitRank")
        h = _NeverInit(rank, rccl_fails=False)
    else:                        # partial failure: rank 1's communicator does not come up, rank 0's does
        def make_id():
            return bytes(range(128))
        h = _FakeCommHandle(rank, rccl_fails=(rank == 1))
    kind = crl_dist.attach_comm(dist, h, world, rank, "rccl", make_id, fallback=True)
    assert kind.startswith("peer") and "failed" in kind, kind
    assert h.attached is not None and len(h.attached) == 64 * world and not h.comm_up
    if case == "partial":
        assert h.destroyed == (1 if rank == 0 else 0)
    # without fallback every rank raises (and none hangs)
    h2 = _FakeCommHandle(rank, rccl_fails=(rank == 1) if case == "partial" else False) if case != "no_lib_rank1" else type(h)(rank, rccl_fails=False)
    try:
        crl_dist.attach_comm(dist, h2, world, rank, "rccl", make_id, fallback=False)
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    open(os.path.join(out_dir, f"{case}{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["no_id", "partial", "no_lib_rank1"])
def test_rccl_start_failures_fall_back_to_the_peer_allreduce_on_every_rank(tmp_path, case):
    """attach_comm(fallback=True): (1) rank 0 cannot create the id — the other ranks must not block in the broadcast; (2) RCCL comes
    up on some ranks only — those drop their communicator before exporting a mailbox. Both end with every rank on the peer path."""
    world = 2
    mp.spawn(_fallback_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / f"{case}{r}").exists()
