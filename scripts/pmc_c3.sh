# PMC passes over the layer-wise path's GEMM kernels (bench.py --workload c3): where the 256-wide layers lose their time.
#   bash scripts/pmc_c3.sh <tag>  → gpurun_out/<tag>_c3pmc/*.csv ; each pass its own rocprofv3 run with --pmc only
TAG=${1:-c3}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${TAG}_c3pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline"
KR="wide_dense_x2_kernel|wide_wgrad_x2_kernel|wide_skinny_kernel|wide_dense_kernel"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$KR" --output-format csv -d $O/p_$n -- $P > /dev/null 2>&1
  for f in $(find $O/p_$n -name "*counter_collection.csv"); do cp $f $O/${n}.csv; done
  rm -rf $O/p_$n
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-30s %.4g  (%d launches)" % (n, sum(v) / len(v), len(v)))
PY
