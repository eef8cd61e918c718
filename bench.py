#!/usr/bin/env python3
"""bench.py — env-steps/sec of the whole PPO loop (rollout + GAE + update; ppo.jl:117-253) on MI355X.

A "step" is one PPO iteration over one batch of synthetic CartPole rollouts: num_steps=128 env steps of every env,
GAE, then update_epochs=4 x num_minibatches=4 optimiser steps. Workload (BASELINE.json metric): CartPole PPO,
num_envs=65536 in total, 2x64 actor/critic, reference default hyper-parameters. With N GPUs the envs are sharded
65536/N per rank (strong scaling); per optimiser step ONE RCCL all-reduce carries the flat gradient + 4 loss sums, and per
iteration one more carries the advantage sums of all 16 minibatches.

  python bench.py --gpus N --steps K --warmup W
      N = 1: runs in this process.
      N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N … bench.py --gpus N …` (one rank per
             GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the env), or as the bare command above — then this process,
             which never touches the GPU, starts exactly that launcher as a child and exits with its code.
  --dry-run           print the launcher command and the per-rank environment instead of running (works without a GPU)
  --rendezvous-only   ranks meet over gloo, exchange a communicator id and exit (CPU test of the spawn + id exchange)
  --minibatches 1     north_star's "single all-reduce per update epoch" (one optimiser step per epoch, ppo.jl:5)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

TOTAL_ENVS = 65536
NUM_STEPS = 128
FWD_FLOPS_PER_SAMPLE = 17792          # actor + critic forward, 2x64, obs 4, act 2
GAE_BYTES_PER_STEP, GAE_BYTES_PER_ENV = 17, 5
UPDATE_BYTES_PER_SAMPLE = 36          # obs 16 + action 4 + old logprob 4 + adv 4 + return 4 + old value 4
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA
PEAK_HBM_GBPS = 8000.0
PROFILE_TAG = "r02"                   # profiles/<tag>_* hold the rocprofv3 summaries of this round


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def physical_cores():
    """(physical cores, logical cpus) of this box from /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        return (len(cores) or logical), logical
    except OSError:
        return logical, logical


def _set_omp_threads(n):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _time_oracle(nt, threads, budget_s, min_iters=1):
    import oraclelib as O
    _set_omp_threads(threads)
    cfg = O.make_config(num_envs=nt, num_steps=NUM_STEPS)
    st = O.State(cfg)
    st.params[:] = O.orthogonal_params(cfg, 0)
    st.env_init()
    t0 = time.perf_counter()
    st.iterate(1000)  # warm-up (page-in, OpenMP pool); also sizes the timed part
    warm = time.perf_counter() - t0
    iters_goal = max(min_iters, int(budget_s / max(warm, 1e-3)))
    t0 = time.perf_counter(); iters = 0
    while iters < iters_goal:
        st.iterate(1000); iters += 1
    dt = time.perf_counter() - t0
    st.close()
    return nt * NUM_STEPS * iters / dt, iters, dt


def cpu_baseline():
    """The CPU restatement of ppo.jl (oracle/ppo_oracle.c, kind="port": scalar C, -O2 -ffp-contract=off, OpenMP over envs /
    samples) timed on this box's host cores on bounded samples of the same loop: nt=4096 on all physical cores
    (the headline `value`), C1 (nt=8) single-threaded, and C1 on all cores."""
    cores, logical = physical_cores()
    v_all, it_all, dt_all = _time_oracle(4096, cores, budget_s=12.0)
    v_1, it_1, dt_1 = _time_oracle(8, 1, budget_s=4.0, min_iters=3)
    v_c1, it_c1, dt_c1 = _time_oracle(8, min(cores, 8), budget_s=2.0, min_iters=3)
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"num_envs=4096, num_steps={NUM_STEPS}, {it_all} full PPO iterations (rollout+GAE+16 optimiser steps) in {dt_all:.1f}s, "
                      f"OpenMP threads = {cores} physical cores ({logical} logical cpus)",
            "single_thread": {"value": v_1, "unit": "env-steps/s", "cores": 1,
                              "sample": f"C1: num_envs=8, num_steps={NUM_STEPS}, {it_1} iterations in {dt_1:.1f}s"},
            "c1_multi_thread": {"value": v_c1, "unit": "env-steps/s", "cores": min(cores, 8),
                                "sample": f"C1: num_envs=8 (one env per thread), {it_c1} iterations in {dt_c1:.1f}s"}}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launcher_command(args, argv, port):
    """The command the bare `bench.py --gpus N` runs as a child: one fresh process per GPU. Nothing in THIS process has
    touched HIP (no torch import), so no GPU-initialised process is ever replaced or forked."""
    rest = [a for a in argv if a not in ("--dry-run",)]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def rank_env(world, rank, port):
    return {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
            "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_SOCKET_IFNAME": "lo"}


def spawn_ranks(args, argv):
    port = args.master_port or _free_port()
    cmd = launcher_command(args, argv, port)
    if args.dry_run:
        print(json.dumps({"launcher": cmd}))
        for r in range(args.gpus):
            print(json.dumps({"rank": r, "cmd": [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-run"],
                              "env": rank_env(args.gpus, r, port)}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-envs", type=int, default=TOTAL_ENVS)
    ap.add_argument("--minibatches", type=int, default=4, help="num_minibatches (ppo.jl:5); 1 = one optimiser step and one gradient all-reduce per epoch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-breakdown", action="store_true",
                    help="time every kernel class with recorded HIP events (kernel_ms_per_step gets all entries; the events cost "
                         "about 0.3 ms per iteration, so `value` is a little lower than in the default run)")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--rendezvous-only", action="store_true")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--comm", choices=["rccl", "peer"], default="rccl",
                    help="gradient exchange for --gpus > 1: rccl = ncclAllReduce (default); peer = the one-shot peer-mapped "
                         "all-reduce of csrc/peer.hip (crl_comm_peer_export/attach)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="functional check on a 1-GPU box: all ranks run on GPU 0 (use --comm peer; RCCL refuses to put two ranks on "
                         "one device and the run falls back to the peer all-reduce). The line is labelled shared_gpu and is not a scaling measurement")
    ap.add_argument("--shuffle", choices=["bijection", "fisher-yates", "blocked-fy"], default="blocked-fy",
                    help="blocked-fy = exact parallel Fisher-Yates (uniform over S_B like the reference shuffle; default); "
                         "bijection = keyed pseudo-random permutation (faster, not a uniform draw); fisher-yates = serial exact")
    ap.add_argument("--workload", choices=["cartpole", "c3"], default="cartpole",
                    help="cartpole = the headline workload (BASELINE metric); c3 = BASELINE configs[2]: LunarLander-shaped obs 8 / "
                         "act 4, 2x256 MLP, num_envs=16384 on the synthetic env (a side measurement, not the driver's line)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run or (world == 1 and args.gpus > 1):
        if args.gpus < 2:
            print(json.dumps({"launcher": None, "note": "--gpus 1 runs in-process"}))
            return 0
        return spawn_ranks(args, sys.argv[1:])

    c3 = args.workload == "c3"
    if c3 and args.total_envs == TOTAL_ENVS:
        args.total_envs = 16384
    fwd_flops = 272896 if c3 else FWD_FLOPS_PER_SAMPLE
    upd_flops = 3 * fwd_flops                             # forward + backward (≈2x forward) per sample per optimiser pass
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if args.share_gpu:
        local_rank = 0   # with --comm rccl this exercises the fallback: RCCL refuses two ranks on one device

    # stdout carries exactly ONE JSON line (rank 0). Native libraries (RCCL prints a version banner through C stdio,
    # flushed at exit) must not leak into it: fd 1 is pointed at stderr for the whole run and the JSON goes to the saved fd.
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import importlib
    import torch
    import torch.distributed as dist
    crl_dist = importlib.import_module("cleanrl_jl_amd.dist")

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL's bootstrap must not depend on the hostname resolving
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # control plane; gradients go over RCCL
    if args.rendezvous_only:
        uid = crl_dist.exchange_unique_id(dist, rank, lambda: bytes((7 * i + 3) % 256 for i in range(128))) if world > 1 else b""
        nt_local, env_off = crl_dist.shard_envs(args.total_envs, world, rank)
        ok = torch.tensor([1.0 if (world == 1 or uid[5] == 38) else 0.0])
        if world > 1:
            dist.all_reduce(ok)
            dist.barrier()
        if rank == 0:
            os.write(json_fd, (json.dumps({"rendezvous": "ok" if int(ok.item()) == world else "failed", "world": world,
                                           "envs_per_rank": nt_local}) + "\n").encode())
        if world > 1:
            dist.destroy_process_group()
        return 0

    import cleanrl_jl_amd as crl
    L = crl._lib
    torch.cuda.set_device(local_rank)

    nt_local, env_off = crl_dist.shard_envs(args.total_envs, world, rank)
    cfg = crl.PPOConfig(num_envs=nt_local, num_steps=NUM_STEPS, num_minibatches=args.minibatches,
                        total_timesteps=args.total_envs * NUM_STEPS * (args.steps + args.warmup + 1))
    shape = dict(obs_dim=8, n_act=4, hidden=256, env_kind=L.ENV_SYNTHETIC) if c3 else {}
    agent = crl.Agent(cfg, device=local_rank, env_id_offset=env_off, **shape,
                      shuffle_mode={"bijection": L.SHUFFLE_BIJECTION, "fisher-yates": L.SHUFFLE_FISHER_YATES,
                                    "blocked-fy": L.SHUFFLE_BLOCKED_FY}[args.shuffle])
    h = agent.handle
    comm_used = args.comm
    if world > 1:
        comm_used = crl_dist.attach_comm(dist, h, world, rank, args.comm, crl.comm_unique_id, fallback=True)
    elif os.environ.get("CRL_COMM_FORCE") and args.comm == "peer":
        h.comm_peer_attach(h.comm_peer_export(1, 0))   # 1-rank mailbox: the all-reduce kernel still runs (push to self)
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
    # level 2 = only the update kernel, whose events ride on the dispatch (no extra packets in the timed stream); --kernel-breakdown
    # (and the layer-wise workload, whose optimiser pass is a group of launches) records events around every kernel class instead
    h.prof_enable(1 if (args.kernel_breakdown or c3) else 2); h.prof_reset()
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
    reruns = h.exact_reruns

    # GAE alone, cold (caches flushed by a 1 GiB fill first) and warm (launched again right away): the in-loop launch above
    # runs right behind the rollout that produced its inputs, so part of what it reads is still cache-resident.
    gae_bytes = GAE_BYTES_PER_STEP * nt_local * NUM_STEPS + GAE_BYTES_PER_ENV * nt_local
    gae_cold = gae_warm = None
    if rank == 0 and not c3 and world == 1:
        flush = torch.empty(1 << 28, dtype=torch.float32, device=f"cuda:{local_rank}")
        cold, warm = [], []
        for i in range(6):
            flush.fill_(float(i)); torch.cuda.synchronize()
            h.prof_enable(True); h.prof_reset(); h.compute_gae(); h.sync()
            cold.append(h.prof_read()["gae"][0])
            h.prof_reset(); h.compute_gae(); h.sync()
            warm.append(h.prof_read()["gae"][0]); h.prof_enable(False)
        del flush
        cold.sort(); warm.sort()
        gae_cold, gae_warm = cold[len(cold) // 2], warm[len(warm) // 2]
    agent.close()

    if rank == 0:
        env_steps = args.total_envs * NUM_STEPS * args.steps
        M = nt_local * NUM_STEPS // cfg.num_minibatches
        upd_ms, upd_n = prof["update"]
        gae_ms, gae_n = prof["gae"]
        upd_avg_s = upd_ms / max(upd_n, 1) * 1e-3
        gae_avg_s = gae_ms / max(gae_n, 1) * 1e-3
        upd_tflops = upd_flops * M / upd_avg_s / 1e12 if upd_n else 0.0
        gae_gbps = gae_bytes / gae_avg_s / 1e9 if gae_n else 0.0
        gemm = os.environ.get("CRL_GEMM", "x2")
        x3 = gemm != "f32"
        # matrix-pipe products issued per f32 product of the three hidden-layer GEMMs (forward, backward-data, weight gradient):
        # x2 = fp16x2: 3 (a tile whose cotangents fall outside the launch's scale window takes bf16x3 for its weight gradient: 6);
        # x3 = bf16x3 everywhere: 6; C3 (wide.hip) runs bf16x3
        issue_factor = 1.0 if not x3 else (6.0 if (gemm == "x3" or c3) else 3.0)
        # share of the algorithmic flops that runs as 64x64 (256x256) products = what goes to the matrix pipe
        hh = 2 * 2 * 64 * 64 if not c3 else 2 * 2 * 256 * 256
        mfma_share = hh / fwd_flops
        # HBM traffic per launch from committed rocprofv3 PMC passes of this same command (FETCH_SIZE and WRITE_SIZE in
        # separate passes, scripts/final_measure.sh): NOT measured inside this run — `traffic_source` names the file.
        traffic = {"update": None, "gae": None}
        traffic_file = os.path.join("profiles", f"{PROFILE_TAG}_pmc_hbm_traffic.json")
        try:
            pm = json.load(open(os.path.join(ROOT, traffic_file)))
            if world == 1 and args.total_envs == TOTAL_ENVS and not c3 and args.minibatches == 4:
                for key, frag in (("update", "update_"), ("gae", "gae_kernel")):
                    for name, rec in pm.items():
                        if frag in name and "vfix" not in name:
                            # how FETCH_SIZE compares with bytes for this kernel's access pattern (scripts/summarize_pmc.py): 2 for
                            # 16-B-per-lane streaming reads (the guide's gfx950 correction), 1 for gathered 64-byte records
                            fx = float(rec.get("fetch_factor", 2.0 if rec.get("fetch_x2_corrected") else 1.0))
                            traffic[key] = (fx * rec["FETCH_SIZE_KB_per_launch_mean"] + rec["WRITE_SIZE_KB_per_launch_mean"]) * 1024
                            break
        except Exception:
            traffic_file = None
        workload = (f"PPO CartPole-v1-shaped on-device env, num_envs={args.total_envs} total ({nt_local}/GPU), num_steps={NUM_STEPS}, "
                    f"2x64 actor+critic MLP, update_epochs=4, num_minibatches={cfg.num_minibatches}, anneal_lr") if not c3 else \
                   (f"PPO LunarLander-shaped synthetic env (obs 8, act 4), num_envs={args.total_envs} total ({nt_local}/GPU), "
                    f"num_steps={NUM_STEPS}, 2x256 actor+critic MLP, update_epochs=4, num_minibatches={cfg.num_minibatches}, anneal_lr")
        out = {
            "metric": "env-steps/sec (whole node), CartPole PPO num_envs=65536 at 1/2/4/8 GPUs" if not c3 else
                      "env-steps/sec, PPO LunarLander-shaped (obs 8 / act 4, 2x256) num_envs=16384 (BASELINE configs[2], side measurement)",
            "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": args.total_envs * NUM_STEPS, "parallelism": f"dp{world}",
                       "comm": (None if world == 1 and not os.environ.get("CRL_COMM_FORCE") else
                                "rccl all-reduce" if comm_used == "rccl" else
                                "one-shot peer-mapped all-reduce (csrc/peer.hip)" + (" — RCCL initialisation failed" if "failed" in comm_used else "")),
                       **({"shared_gpu": f"all {world} ranks time-share GPU 0 (functional check of the multi-rank path, NOT a scaling "
                                         "measurement)"} if args.share_gpu and world > 1 else {}),
                       "shuffle": args.shuffle,
                       "gemm": ("v_mfma_f32_32x32x2_f32" if not x3 else
                                "f32 results via bf16x3 split products on the bf16 matrix pipe" if issue_factor == 6.0 else
                                "f32 results via fp16x2 split products (3 per f32 product) on the f16 matrix pipe")},
            "roofline": {"bound": "mfma",
                         "kernel": "update kernel (fwd+bwd of one minibatch, actor+critic)" if not c3 else
                                   "wide.hip: all forward/backward launches of one minibatch (HIP events around the group)",
                         "achieved": upd_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": upd_tflops / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic["update"], "traffic_source": traffic_file if traffic["update"] else None,
                         "algorithmic_bytes_per_launch": UPDATE_BYTES_PER_SAMPLE * M if not c3 else None,
                         "avg_launch_ms": upd_avg_s * 1e3, "launches": upd_n, "flops_per_launch": upd_flops * M,
                         "pipe": {"name": "f16/bf16 mfma" if x3 else "f32 mfma",
                                  "issued_tflops": upd_tflops * mfma_share * issue_factor,
                                  "peak": PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS,
                                  "frac": upd_tflops * mfma_share * issue_factor / (PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS)},
                         "note": f"achieved = ALGORITHMIC f32 flops (3 x {fwd_flops:,} per sample) / HIP-event launch time, against the dense "
                                 "f32-MFMA peak (SURVEY 8d's denominator): an f32-equivalent figure, which can exceed 1 because that pipe is not the one used. `pipe` prices what is actually issued: the hidden-layer products "
                                 f"({mfma_share:.0%} of the flops) run as {issue_factor:g} f16/bf16 partial products per f32 product on the matrix pipe, so "
                                 "pipe.frac is that pipe's utilisation; the rest (tanh, splits, loss, skinny gradients) is VALU work — see "
                                 f"profiles/{PROFILE_TAG}_*pmc* for the measured issue/wait split"},
            "roofline_gae": {"bound": "hbm", "kernel": "gae_kernel (advantages + returns), standalone launch (crl_compute_gae)",
                             "peak": PEAK_HBM_GBPS, "unit": "GB/s", "traffic": traffic["gae"],
                             "traffic_source": traffic_file if traffic["gae"] else None, "bytes_per_launch": gae_bytes,
                             "in_loop": ({"launches": gae_n, "avg_launch_ms": gae_avg_s * 1e3, "achieved": gae_gbps, "frac": gae_gbps / PEAK_HBM_GBPS,
                                          "note": "timed inside the iteration, right behind the rollout (inputs partly cache-resident)"} if gae_n else
                                         {"launches": 0, "note": "inside crl_ppo_iterate the compat-mode scan is fused into the tail of the rollout "
                                                                 "kernel (each wave scans the 32 envs it just stepped, inputs still in L2): no launch, no "
                                                                 "HBM read of the scan's inputs; the figures here are the standalone kernel's"})},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if (args.kernel_breakdown or c3 or v[1] > 0)},
            "kernel_ms_scope": ("every kernel class (events recorded around the launches; they cost about 0.3 ms per iteration)"
                                if (args.kernel_breakdown or c3) else
                                "update kernel only (events attached to its dispatch: no extra packets in the timed stream); "
                                "--kernel-breakdown times every class"),
            "last_iteration": {"loss": stats[-1]["loss"], "episodes": ep["episodes"],
                               "mean_episode_return": ep["return_sum"] / max(ep["episodes"], 1.0), "exact_reruns": reruns},
        }
        if gae_cold is not None:
            for name, ms in (("cold", gae_cold), ("warm", gae_warm)):
                out["roofline_gae"][name] = {"avg_launch_ms": ms, "achieved": gae_bytes / (ms * 1e-3) / 1e9,
                                             "frac": gae_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}
            # headline figure of the GAE roofline: the standalone launch on cache-cold inputs (caches flushed by a 1 GiB fill)
            out["roofline_gae"].update({"achieved": out["roofline_gae"]["cold"]["achieved"], "frac": out["roofline_gae"]["cold"]["frac"],
                                        "avg_launch_ms": gae_cold, "state": "cold"})
        elif gae_n:
            out["roofline_gae"].update({"achieved": gae_gbps, "frac": gae_gbps / PEAK_HBM_GBPS, "avg_launch_ms": gae_avg_s * 1e3, "state": "in-loop"})
        if world == 1 and not args.no_cpu_baseline and not c3:
            out["cpu_baseline"] = cpu_baseline()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
