#!/bin/bash
# SQ counters of the attention kernel alone (tools/attn_bench.bin, ViT-L/16-384 shape by default).  usage: tools/pmc_attn.sh <outdir> [bench args...]
OUT=${1:-gpurun_out/pmc_attn}; shift; ARGS=${@:-128 577 16 64 2}; mkdir -p $OUT; export TMPDIR=/tmp
BIN=${ATTN_BIN:-tools/attn_bench.bin}
run() { rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT/$2 -- $BIN $ARGS > $OUT/$2.log 2>&1 || echo "pass $2 failed"; }
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" sq
run "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" sq2
run "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32" sq3
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):18.1f}")
PY
