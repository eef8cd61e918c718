"""Rank body of tests/test_gpu_dp2.py: one process per rank, launched by torch.distributed.run before anything touches HIP.
DP2_COMM = rccl (one GPU per rank) or peer (the one-shot peer-mapped all-reduce; DP2_SHARE_GPU=1 puts every rank on GPU 0 so a
1-GPU box runs the real multi-process path). Checks with world size = number of ranks:
  1. host-driven optimiser steps on shards (local permutations, global advantage statistics through the library's all-reduce)
     == the same steps of ONE handle that owns the union batch (rank 0 checks, 1e-5 relative per parameter array);
  2. crl_ppo_iterate keeps the replicas bit-identical (all-reduced gradients are the same bytes everywhere);
  3. a failed value-loss speculation (gamma = 0, critic bias 5) is repaired by the guard window on every rank, replicas stay
     bit-identical and exact_reruns counts the repeated iterations."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import importlib
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cleanrl_jl_amd as crl
    crl_dist = importlib.import_module("cleanrl_jl_amd.dist")
    L = crl._lib
    kind = os.environ.get("DP2_COMM", "rccl")
    if os.environ.get("DP2_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)

    def attach(hh):
        hh.set_option("peer_timeout_ms", int(os.environ.get("DP2_PEER_TIMEOUT_MS", "60000")))
        hh.set_option("fuse_optim", int(os.environ.get("DP2_FUSE_OPTIM", "1")))   # 1: reduce + peer exchange + ClipNorm + Adam as ONE launch inside crl_ppo_iterate
        crl_dist.attach_comm(dist, hh, world, rank, kind, crl.comm_unique_id)
    NT, k = 16 * world, 128
    n, off = crl_dist.shard_envs(NT, world, rank)
    out = {}

    if os.environ.get("DP2_MODE") == "timeout":
        # a rank that stops answering: rank 0 issues an all-reduce nobody else joins — its kernel must give up after
        # option peer_timeout_ms and the next synchronising call must report it (no hang, no garbage passed on silently)
        cfg = crl.PPOConfig(num_envs=n, num_steps=k, total_timesteps=NT * k * 10)
        a = crl.Agent(cfg, device=local, env_id_offset=off, init_seed=3)
        hh = a.handle
        attach(hh)
        hh.env_reset(); hh.rollout_run(); hh.compute_gae()
        msg = ""
        if rank == 0:
            hh.write(L.F_PERM, np.arange(n * k, dtype=np.int32))
            try:
                hh.adv_stats()        # local sums -> all-reduce (alone) -> mean / std
                hh.sync()
            except crl.CrlError as e:
                msg = str(e)
            print("DP2_RESULT " + json.dumps({"timeout_error": msg}), flush=True)
        dist.barrier()               # rank 1 keeps its mailbox mapped until rank 0 is through
        a.close()
        dist.barrier()
        dist.destroy_process_group()
        return

    def gather(a):
        parts = [None] * world
        dist.all_gather_object(parts, np.ascontiguousarray(a))
        return parts

    # ---- 1. host-driven steps vs the union handle
    cfg = crl.PPOConfig(num_envs=n, num_steps=k, total_timesteps=NT * k * 10)
    shard = crl.Agent(cfg, device=local, env_id_offset=off, init_seed=3)
    h = shard.handle
    attach(h)
    h.env_reset(); h.rollout_run(); h.compute_gae()
    perm = np.random.default_rng(100 + rank).permutation(n * k).astype(np.int32)
    h.write(L.F_PERM, perm)
    h.adv_stats()                                   # local sums -> RCCL all-reduce -> global mean / std
    for mb in range(4):
        h.update_minibatch(mb, 2.5e-4, apply_update=True, want_stats=False)
    p_shard = h.read(L.F_PARAMS)
    perms = gather(perm); params_all = gather(p_shard)
    out["replicas_equal_after_steps"] = all(np.array_equal(params_all[0], q) for q in params_all)
    if rank == 0:
        full = crl.Agent(crl.PPOConfig(num_envs=NT, num_steps=k, total_timesteps=NT * k * 10), device=local, init_seed=3)
        hf = full.handle
        hf.env_reset(); hf.rollout_run(); hf.compute_gae()
        Ml = n * k // 4
        union = []
        for mb in range(4):
            for r in range(world):
                loc = perms[r][mb * Ml:(mb + 1) * Ml]
                union.append((r * n + loc % n) + NT * (loc // n))
        hf.write(L.F_PERM, np.concatenate(union).astype(np.int32))
        hf.adv_stats()
        for mb in range(4):
            hf.update_minibatch(mb, 2.5e-4, apply_update=True, want_stats=False)
        p_full = hf.read(L.F_PARAMS)
        out["max_abs_param_diff_vs_union"] = float(np.max(np.abs(p_full - p_shard)))
        out["rel_l2_vs_union"] = float(np.linalg.norm(p_full.astype(np.float64) - p_shard) / np.linalg.norm(p_full))
        full.close()
    dist.barrier()
    shard.close()

    # ---- 2. + 3. whole iterations; second run forces the u > q branch
    for name, gamma, bias in (("iterate", 0.99, None), ("forced_branch", 0.0, 5.0)):
        cfg = crl.PPOConfig(num_envs=n, num_steps=k, total_timesteps=NT * k * 10, gamma=gamma)
        a = crl.Agent(cfg, device=local, env_id_offset=off, init_seed=4)
        if bias is not None:
            p = a.get_params(); p[-1] = bias; a.set_params(p)        # critic head bias is the last parameter
        hh = a.handle
        attach(hh)
        hh.env_reset()
        hh.iterate(2, want_stats=False)
        stats = hh.iterate(1)
        ps = gather(hh.read(L.F_PARAMS))
        losses = gather(np.array([[st[key] for key in ("loss", "pg_loss", "v_loss", "entropy_loss")] for st in stats]))
        ms = gather(hh.read(L.F_ADAM_M)); gr = gather(hh.read(L.F_GRADS))
        out[name] = {"replicas_equal": all(np.array_equal(ps[0], q) for q in ps) and all(np.array_equal(ms[0], q) for q in ms),
                     "stats_equal": all(np.array_equal(losses[0], q) for q in losses) and bool(np.isfinite(losses[0]).all()),
                     "grads_equal": all(np.array_equal(gr[0], q) for q in gr),
                     "exact_reruns": hh.exact_reruns, "loss": stats[-1]["loss"], "finite": bool(np.isfinite(ps[0]).all()),
                     "params_l2": float(np.linalg.norm(ps[0].astype(np.float64))), "fuse_optim": hh.get_option("fuse_optim"),
                     "grads_l2": float(np.linalg.norm(gr[0].astype(np.float64))), "first_step": [float(x) for x in losses[0][0]], "last_step": [float(x) for x in losses[0][-1]]}
        dist.barrier()   # nobody frees a mailbox a slower rank could still be pushing into
        a.close()
    if rank == 0:
        print("DP2_RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
