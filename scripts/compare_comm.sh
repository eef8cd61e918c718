# On a multi-GPU MI355X node: the data-parallel bench with both gradient exchanges (RCCL and the one-shot peer-mapped all-reduce of
# csrc/peer.hip) at N = 2, 4, 8 — the measurement this project's 1-GPU boxes cannot make. Each run is bench.py's JSON line with
# kernel_ms_per_step.allreduce (16 gradient messages + 1 advantage-sum message per iteration); the summary of all runs goes to
# <out_dir>/summary.json (one object: {"n2_rccl": {...}, "n2_peer": {...}, …}).
#   bash scripts/compare_comm.sh [out_dir]
O=${1:-gpurun_out/compare_comm}
mkdir -p $O
G=$(python3 -c "import torch; print(torch.cuda.device_count())")
for n in 2 4 8; do
  [ "$n" -le "$G" ] || continue
  for comm in rccl peer; do
    timeout 900 python bench.py --gpus $n --comm $comm --no-cpu-baseline 2>$O/n${n}_$comm.err | grep '^{' > $O/n${n}_$comm.json
  done
done
python3 - "$O" <<'PY'
import glob, json, os, sys
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "n*_*.json"))):
    key = os.path.basename(f)[:-5]
    try:
        d = json.load(open(f))
        out[key] = {"n_gpus": d["n_gpus"], "value": d["value"], "ms_per_step": d["ms_per_step"], "comm": d["config"].get("comm"),
                    "allreduce_ms_per_step": d["kernel_ms_per_step"].get("allreduce"), "update_ms_per_launch": d["roofline"]["avg_launch_ms"]}
    except Exception as e:   # noqa: BLE001
        out[key] = {"error": str(e)}
json.dump(out, open(os.path.join(sys.argv[1], "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
