"""Where does the standalone GAE kernel's WRITE_SIZE above its 67.1 MB of outputs come from (103.3 MB = 1.54 x)?
Hypothesis: dirty L2 lines of the PREDECESSOR kernel that are evicted while the scan runs are tallied to the scan's dispatch. This script
runs, at 65536 envs x 128, one rollout followed by FIVE back-to-back crl_compute_gae launches, then a 1 GiB fill and one more:
    rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gae_kernel" --output-format csv -d <dir> -- python3 scripts/pmc_gae_write.py
If the hypothesis holds, the first launch (behind the rollout, which leaves its last stores dirty in L2) and the one behind the fill
report more than 67.1 MB and the back-to-back ones (behind a predecessor whose leftovers they overwrite with lines of their own) about 67.1 MB."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import cleanrl_jl_amd as crl  # noqa: E402

agent = crl.Agent(crl.PPOConfig(num_envs=65536, num_steps=128, total_timesteps=65536 * 128 * 10))
h = agent.handle
h.env_reset(); h.rollout_run()
for _ in range(5):
    h.compute_gae()
h.sync()
if len(sys.argv) > 1 and sys.argv[1] == "fill":      # round 5's evidence run (profiles/r05_gae_write_size.txt): the segmented kernel behind a 1 GiB fill each time
    g, c = crl._lib.gae_bench(65536, 128, seg=8, tile=64, nt_loads=0, flush_mb=1024, reps=2)
h.rollout_run(); h.compute_gae(); h.sync()
agent.close()
print("done")
