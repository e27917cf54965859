#!/bin/bash
# PMC passes over the GEMM microbenchmark (one shape, chosen variants); counters in separate runs.
# usage: tools/pmc_gemm.sh <shape_idx> <variant_mask> <outdir>
set -e
SHAPE=${1:-0}; MASK=${2:-0xa}; OUT=${3:-gpurun_out/pmc_gemm}
mkdir -p $OUT
export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc "$1" --output-format csv -d $OUT/$2 -- tools/gemm_bench.bin 12608 $SHAPE $MASK 1 > $OUT/$2.log 2>&1 || echo "pass $2 failed"; }
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" sq
run "TCC_HIT_sum TCC_MISS_sum" tcc
run "FETCH_SIZE" fetch
run "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" sq2
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
