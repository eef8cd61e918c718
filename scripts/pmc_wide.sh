mkdir -p gpurun_out/pmcw; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "wide_dense_x3_kernel<., 2|wide_wgrad_x3" --output-format csv -d $R/gpurun_out/pmcw/p1 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 > $R/gpurun_out/pmcw/p1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE --kernel-include-regex "wide_dense_x3_kernel<., 2|wide_wgrad_x3" --output-format csv -d $R/gpurun_out/pmcw/p2 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 > $R/gpurun_out/pmcw/p2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES --kernel-include-regex "wide_dense_x3_kernel<., 2|wide_wgrad_x3" --output-format csv -d $R/gpurun_out/pmcw/p3 -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 > $R/gpurun_out/pmcw/p3.log 2>&1
cd $R/gpurun_out/pmcw; for p in p1 p2 p3; do f=$(find $p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $p.csv; done; rm -rf p1 p2 p3
echo done
