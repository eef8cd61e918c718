mkdir -p gpurun_out/pmcl; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-include-regex "update_x3|rollout|bfy_leaf|bfy_l1" --output-format csv -d $R/gpurun_out/pmcl/p1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcl/p1.log 2>&1
cd $R/gpurun_out/pmcl; f=$(find p1 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f p1.csv; rm -rf p1
