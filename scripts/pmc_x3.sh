mkdir -p gpurun_out/pmcx; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-include-regex "update_x3|rollout" --output-format csv -d $R/gpurun_out/pmcx/p1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcx/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-include-regex "update_x3|rollout" --output-format csv -d $R/gpurun_out/pmcx/p2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcx/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-include-regex "update_x3|rollout" --output-format csv -d $R/gpurun_out/pmcx/p3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcx/p3.log 2>&1
cd $R/gpurun_out/pmcx; for p in p1 p2 p3; do f=$(find $p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $p.csv; done; rm -rf p1 p2 p3
echo done
