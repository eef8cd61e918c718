# On a multi-GPU MI355X node: the data-parallel bench with both gradient exchanges (RCCL and the one-shot peer-mapped all-reduce of
# csrc/peer.hip) at N = 2, 4, 8 — the measurement this project's 1-GPU boxes cannot make. Each line is bench.py's JSON with
# kernel_ms_per_step.allreduce (16 gradient messages + 1 advantage-sum message per iteration).
#   bash scripts/compare_comm.sh [out_dir]
O=${1:-gpurun_out/compare_comm}
mkdir -p $O
G=$(python3 -c "import torch; print(torch.cuda.device_count())")
for n in 2 4 8; do
  [ "$n" -le "$G" ] || continue
  for comm in rccl peer; do
    timeout 900 python bench.py --gpus $n --comm $comm --no-cpu-baseline 2>$O/n${n}_$comm.err | grep '^{' > $O/n${n}_$comm.json
    python3 - "$O/n${n}_$comm.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "n_gpus", d["n_gpus"], "%.3f ms/iter" % d["ms_per_step"], "%.4g env-steps/s" % d["value"], d["config"].get("comm"),
          "allreduce ms/iter %.3f" % d["kernel_ms_per_step"].get("allreduce", float("nan")))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  done
done
