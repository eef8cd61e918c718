#!/usr/bin/env python3
"""bench.py — env-steps/sec of the whole PPO loop (rollout + GAE + update; ppo.jl:117-253) on MI355X.

A "step" is one PPO iteration over one batch of synthetic CartPole rollouts: num_steps=128 env steps of every env,
GAE, then update_epochs=4 x num_minibatches=4 optimiser steps. Workload (BASELINE.json metric): CartPole PPO,
num_envs=65536 in total, 2x64 actor/critic, reference default hyper-parameters. With N GPUs the envs are sharded
65536/N per rank (strong scaling) and the flat gradient is all-reduced over RCCL once per optimiser step.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

TOTAL_ENVS = 65536
NUM_STEPS = 128
FWD_FLOPS_PER_SAMPLE = 17792          # actor + critic forward, 2x64, obs 4, act 2
UPDATE_FLOPS_PER_SAMPLE = 3 * FWD_FLOPS_PER_SAMPLE   # forward + backward (≈2x forward) per sample per optimiser pass
GAE_BYTES_PER_STEP, GAE_BYTES_PER_ENV = 17, 5
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBPS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(target_seconds=12.0):
    """The CPU restatement of ppo.jl (oracle, kind="port") timed on this box's host cores, same workload shape on a
    bounded sample: num_envs=2048 (not 65536), num_steps=128, whole iterations until ~target_seconds have elapsed."""
    import numpy as np
    import oraclelib as O
    nt = 2048
    cfg = O.make_config(num_envs=nt, num_steps=NUM_STEPS)
    st = O.State(cfg)
    st.params[:] = O.orthogonal_params(cfg, 0)
    st.env_init()
    st.iterate(1000)  # warm-up (page-in, OpenMP pool)
    t0 = time.perf_counter(); iters = 0
    while time.perf_counter() - t0 < target_seconds:
        st.iterate(1000); iters += 1
    dt = time.perf_counter() - t0
    st.close()
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": nt * NUM_STEPS * iters / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"num_envs={nt}, num_steps={NUM_STEPS}, {iters} full PPO iterations (rollout+GAE+16 optimiser steps), "
                      f"OpenMP over envs/samples, {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-envs", type=int, default=TOTAL_ENVS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shuffle", choices=["bijection", "fisher-yates", "blocked-fy"], default="blocked-fy",
                    help="blocked-fy = exact parallel Fisher-Yates (uniform over S_B like the reference shuffle; default); "
                         "bijection = keyed pseudo-random permutation (faster, not a uniform draw); fisher-yates = serial exact")
    ap.add_argument("--workload", choices=["cartpole", "c3"], default="cartpole",
                    help="cartpole = the headline workload (BASELINE metric); c3 = BASELINE configs[2]: LunarLander-shaped obs 8 / "
                         "act 4, 2x256 MLP, num_envs=16384 on the synthetic env (a side measurement, not the driver's line)")
    args = ap.parse_args()
    c3 = args.workload == "c3"
    if c3 and args.total_envs == TOTAL_ENVS:
        args.total_envs = 16384
    fwd_flops = 272896 if c3 else FWD_FLOPS_PER_SAMPLE
    upd_flops = 3 * fwd_flops

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    # stdout carries exactly ONE JSON line (rank 0). Native libraries (RCCL prints a version banner through C stdio,
    # flushed at exit) must not leak into it: fd 1 is pointed at stderr for the whole run and the JSON goes to the saved fd.
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import cleanrl_jl_amd as crl
    L = crl._lib

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL's bootstrap must not depend on the hostname resolving
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # control plane; gradients go over RCCL
    torch.cuda.set_device(local_rank)

    import importlib
    crl_dist = importlib.import_module("cleanrl_jl_amd.dist")
    nt_local, env_off = crl_dist.shard_envs(args.total_envs, world, rank)
    cfg = crl.PPOConfig(num_envs=nt_local, num_steps=NUM_STEPS, total_timesteps=args.total_envs * NUM_STEPS * (args.steps + args.warmup))
    shape = dict(obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC) if c3 else {}
    agent = crl.Agent(cfg, device=local_rank, env_id_offset=env_off, **shape,
                      shuffle_mode={"bijection": L.SHUFFLE_BIJECTION, "fisher-yates": L.SHUFFLE_FISHER_YATES,
                                    "blocked-fy": L.SHUFFLE_BLOCKED_FY}[args.shuffle])
    h = agent.handle
    if world > 1:
        h.comm_init(crl_dist.exchange_unique_id(dist, rank, crl.comm_unique_id), world, rank)
    elif os.environ.get("CRL_COMM_FORCE"):
        h.comm_init(crl.comm_unique_id(), 1, 0)  # 1-GPU box: still route the all-reduces through RCCL
    h.env_reset()

    def barrier():
        h.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        h.iterate(1, want_stats=False)
    barrier()
    h.prof_enable(True); h.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.iterate(1, want_stats=False)
    barrier()
    dt = time.perf_counter() - t0
    h.prof_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = h.prof_read()
    ep = h.episode_stats()
    stats = h.iterate(1)  # one extra, untimed, to read the loss records back
    agent.close()

    if rank == 0:
        env_steps = args.total_envs * NUM_STEPS * args.steps
        M = nt_local * NUM_STEPS // cfg.num_minibatches
        upd_ms, upd_n = prof["update"]
        gae_ms, gae_n = prof["gae"]
        upd_avg_s = upd_ms / max(upd_n, 1) * 1e-3
        gae_avg_s = gae_ms / max(gae_n, 1) * 1e-3
        upd_tflops = upd_flops * M / upd_avg_s / 1e12 if upd_n else 0.0
        gae_bytes = GAE_BYTES_PER_STEP * nt_local * NUM_STEPS + GAE_BYTES_PER_ENV * nt_local
        gae_gbps = gae_bytes / gae_avg_s / 1e9 if gae_n else 0.0
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
        # passes of this same command, scripts/final_measure.sh); raw (FETCH+WRITE)*1024 — the guide's x2 FETCH correction
        # is calibrated for 16-B/lane streams only, these kernels use 4-B and gathered 16-B accesses.
        traffic = {"update": None, "gae": None}
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_l_pmc_hbm_traffic.json")))
            if world == 1 and args.total_envs == TOTAL_ENVS and not c3:
                for key, name in (("update", "void crl::update_x3_kernel<4, 2>"), ("gae", "void crl::gae_kernel<32, 8>")):
                    if name in pm:
                        traffic[key] = (pm[name]["FETCH_SIZE_KB_per_launch_mean"] + pm[name]["WRITE_SIZE_KB_per_launch_mean"]) * 1024
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (whole node), CartPole PPO num_envs=65536 at 1/2/4/8 GPUs" if not c3 else
                      "env-steps/sec, PPO LunarLander-shaped (obs 8 / act 4, 2x256) num_envs=16384 (BASELINE configs[2], side measurement)",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"PPO CartPole-v1-shaped on-device env, num_envs={args.total_envs} total "
                                    f"({nt_local}/GPU), num_steps={NUM_STEPS}, 2x64 actor+critic MLP, update_epochs=4, "
                                    f"num_minibatches=4, anneal_lr") if not c3 else
                                   (f"PPO LunarLander-shaped synthetic env (obs 8, act 4), num_envs={args.total_envs} total "
                                    f"({nt_local}/GPU), num_steps={NUM_STEPS}, 2x256 actor+critic MLP, update_epochs=4, "
                                    f"num_minibatches=4, anneal_lr"), "global_batch": args.total_envs * NUM_STEPS,
                       "parallelism": f"dp{world}", "shuffle": args.shuffle,
                       "gemm": "v_mfma_f32_32x32x2_f32 (layer-wise)" if c3 else
                               ("f32 results via bf16x3 split products on the bf16 matrix pipe" if os.environ.get("CRL_GEMM", "x3") != "f32"
                                else "v_mfma_f32_32x32x2_f32")},
            "roofline": {"bound": "mfma", "kernel": "update_x3_kernel / update_kernel (fwd+bwd of one minibatch, actor+critic)" if not c3 else
                                   "wide.hip: all forward/backward launches of one minibatch (HIP events around the group)",
                         "note": f"achieved = algorithmic f32 FLOPs (3 x {fwd_flops:,} per sample) / HIP-event launch time; peak = dense f32 MFMA. "
                                 "The 64x64 / 256x256 products actually run as bf16x3 on the bf16 matrix pipe (6 bf16 MFMA products per f32 "
                                 "product); PMC: that pipe is busy 29 % of SIMD cycles in update_x3_kernel, VALU issue 47 % "
                                 "(profiles/r01_l_pmc_update_rollout.json) - the kernel is VALU/latency-bound, not matrix-pipe-bound",
                         "achieved": upd_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": upd_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic["update"],
                         "avg_launch_ms": upd_avg_s * 1e3, "launches": upd_n, "flops_per_launch": upd_flops * M},
            "roofline_gae": {"bound": "hbm", "kernel": "gae_kernel (advantages + returns)", "achieved": gae_gbps,
                             "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gae_gbps / PEAK_HBM_GBPS, "traffic": traffic["gae"],
                             "avg_launch_ms": gae_avg_s * 1e3, "launches": gae_n, "bytes_per_launch": gae_bytes},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "last_iteration": {"loss": stats[-1]["loss"], "episodes": ep["episodes"],
                               "mean_episode_return": ep["return_sum"] / max(ep["episodes"], 1.0)},
        }
        if world == 1 and not args.no_cpu_baseline and not c3:
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
