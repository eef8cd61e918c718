mkdir -p gpurun_out
out=gpurun_out/variants.txt; : > $out
run() { timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'upd_ms %.4f'%d['roofline']['avg_launch_ms'], 'iter %.3f'%d['ms_per_step'], 'rollout %.3f'%d['kernel_ms_per_step']['rollout'])" >> $out; }
export CRL_UPDATE=split
for rep in 1 2; do
for n in head_slp head_noslp cur_slp cur_noslp; do CRL_LIB_PATH=$PWD/cleanrl.jl_amd/variants/ab_$n.so run "$n"; done
done
echo done
