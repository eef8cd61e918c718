#!/bin/bash
# Round-4 side evidence, one GPU-box call (the headline profile set is scripts/final_measure.sh):
#   before:  bash scripts/build_variant.sh wstamps -DCRL_EXP_WSTAMPS wide ; bash scripts/build_variant.sh stamps -DCRL_EXP_STAMPS update
#            mkdir -p scripts/micro/bin && for m in wstream_rate store_rate; do hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o scripts/micro/bin/$m scripts/micro/$m.hip; done
#   after:   cp gpurun_out/r04e/r04_*.txt profiles/
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
HASH=$(cat cleanrl.jl_amd/csrc/*.hip cleanrl.jl_amd/csrc/*.hpp cleanrl.jl_amd/csrc/*.cpp | sha256sum | cut -c1-12)
hdr() { echo "# $1"; echo "# csrc sha256/12 = $HASH, $(date -u +%Y-%m-%dT%H:%MZ), $(rocminfo 2>/dev/null | grep -m1 -o 'gfx9[0-9a-z]*')"; }
{ hdr "standalone GAE past the Infinity Cache: crl_gae_bench via scripts/bench_gae_big.py"; timeout 300 python scripts/bench_gae_big.py; } > $O/r04_gae_beyond_cache.txt 2>&1
{ hdr "scripts/micro/wstream_rate.hip: streaming 32 KB weight slabs L2 -> LDS, per CU"; timeout 120 scripts/micro/bin/wstream_rate; } > $O/r04_micro_wstream.txt 2>&1
{ hdr "scripts/micro/store_rate.hip: store bandwidth of the chip by access pattern"; timeout 120 scripts/micro/bin/store_rate; } > $O/r04_micro_store.txt 2>&1
{ hdr "in-kernel timelines of the fused 2x256 kernels (variant build -DCRL_EXP_WSTAMPS; s_memrealtime stamps, 100 MHz)"
  export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/wstamps/libcleanrl_hip.so
  echo "== producer/consumer forward (scripts/wstamps_pc.py)"; timeout 200 python scripts/wstamps_pc.py
  echo "== backward (scripts/wstamps_probe.py) — NOTE: this kernel has no registers to spare: the stamped build spills and runs ~2x slower than production (46 vs 23 us per tile);"
  echo "   the proportions are indicative only, the reliable breakdown is scripts/ablate_bwd.sh (timing ablations of the production build)"; timeout 200 python scripts/wstamps_probe.py
  unset CRL_LIB_PATH; } > $O/r04_c3_stamps.txt 2>&1
{ hdr "in-kernel timeline of update_x2_kernel (variant build -DCRL_EXP_STAMPS) at the headline and shard sizes"
  export CRL_LIB_PATH=$R/cleanrl.jl_amd/variants/stamps/libcleanrl_hip.so
  for nt in 65536 8192 4096; do echo "== $nt envs"; timeout 200 python scripts/stamps_probe.py $nt; done
  unset CRL_LIB_PATH; } > $O/r04_update_stamps.txt 2>&1
{ hdr "option update_prio_small at shard sizes, same box (scripts/ab_small.sh)"; bash scripts/ab_small.sh; } > $O/r04_update_prio_small.txt 2>&1
{ hdr "C3 (16384 envs, 8/4/2x256) by pipeline, same box: bench.py --workload c3 --opt wide_fuse=v"
  for v in 3 2 1 0; do timeout 300 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --opt wide_fuse=$v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('wide_fuse=$v  %.4g env-steps/s  %.2f ms/iter  kernels/iter(ms):' % (d['value'], d['ms_per_step']), {k: round(x, 2) for k, x in d['kernel_ms_per_step'].items()}, 'roofline', d['roofline'])"; done
  echo "== rocprofv3 --kernel-trace --stats of the default pipeline"; bash scripts/c3_kernels.sh default --no-extras; } > $O/r04_c3_pipeline.txt 2>&1
for f in $O/r04_*.txt; do echo "--- $f"; tail -n 6 $f; done
