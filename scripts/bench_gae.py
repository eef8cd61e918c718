"""Standalone GAE kernel micro-benchmark (SURVEY §8d sizes): cold / warm launch time and the same-footprint copy ceiling, for a
sweep of the gae_tile / gae_seg options.  python scripts/bench_gae.py [tile,tile,…] [seg,…]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch            # noqa: E402
import bench            # noqa: E402
import cleanrl_jl_amd as crl   # noqa: E402

tiles = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
segs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
torch.cuda.set_device(0)
for nt in (4096, 8192, 16384, 65536):
    for tile in tiles:
        for seg in segs:
            agent = crl.Agent(crl.PPOConfig(num_envs=nt, num_steps=128), options={"gae_tile": tile, "gae_seg": seg})
            h = agent.handle
            h.env_reset(); h.rollout_run()
            for _ in range(3):
                h.compute_gae()
            h.sync()
            r = bench.time_gae_standalone(torch, h, nt, "cuda:0")
            print(json.dumps({"nt": nt, "tile": tile, "seg": seg, "cold_us": round(r["cold"]["avg_launch_ms"] * 1e3, 2), "cold_frac": round(r["cold"]["frac"], 3),
                              "warm_us": round(r["warm"]["avg_launch_ms"] * 1e3, 2), "warm_frac": round(r["warm"]["frac"], 3),
                              "cold_nt_us": round(r["cold_nt_loads"]["avg_launch_ms"] * 1e3, 2), "cold_nt_frac": round(r["cold_nt_loads"]["frac"], 3),
                              "copy_us": round(r["copy_ceiling"]["avg_launch_ms"] * 1e3, 2), "cold_over_copy": round(r["copy_ceiling"]["cold_over_copy"], 3),
                              "cold_nt_over_copy": round(r["copy_ceiling"]["cold_nt_loads_over_copy"], 3)}), flush=True)
            agent.close()
