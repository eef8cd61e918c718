# Ablation timing of the split update kernel (CRL_ABLATE build only; results of ablated runs are garbage, only times matter)
mkdir -p gpurun_out
out=gpurun_out/ablate.txt; : > $out
run() { CRL_UPDATE=split timeout 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'upd_ms %.4f'%d['roofline']['avg_launch_ms'], 'iter %.3f'%d['ms_per_step'])" >> $out; }
CRL_UPDATE=split run "production"
for m in 0 1 2 4 8 16 32 64 82 86 126; do export CRL_DEBUG_ABLATE=$m; run "abl=$m rw=8"; done
export CRL_DEBUG_ABLATE=0 CRL_DEBUG_RW=4; run "abl=0 rw=4"
echo done
