#!/bin/bash
# SQ/TCC counters per kernel over a short bench run (separate passes).  usage: tools/pmc_kernels.sh <outdir>
OUT=${1:-gpurun_out/pmc_k}; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
run() { rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT/$2 -- python3 bench.py $ARGS > $OUT/$2.log 2>&1 || echo "pass $2 failed"; }
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" sq
run "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" sq2
run "TCC_HIT_sum TCC_MISS_sum" tcc
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("ivit::"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
PY
