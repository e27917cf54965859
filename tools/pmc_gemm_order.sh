#!/bin/bash
# HBM-side read traffic (FETCH_SIZE, x2 on gfx950) and L2 hit rate of the 160x128 GEMM under the two tile orders
# (0 = XCD-aware panels, 1 = row bands per XCD), microbenchmark launches.  usage: tools/pmc_gemm_order.sh <outdir>
OUT=${1:-gpurun_out/pmc_order}; mkdir -p $OUT; export TMPDIR=/tmp
for ord in 0 1; do
  for shape in 0 2 3; do
    IVIT_CFGS="0:$ord" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_${ord}_$shape -- ./tools/gemm_bench.bin 12608 $shape 0x2 3 > $OUT/f_${ord}_$shape.log 2>&1
    IVIT_CFGS="0:$ord" rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/t_${ord}_$shape -- ./tools/gemm_bench.bin 12608 $shape 0x2 3 > $OUT/t_${ord}_$shape.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = {0: "qkv 12608x2304x768", 2: "mlp1 12608x3072x768", 3: "mlp2 12608x768x3072"}
for shape in (0, 2, 3):
    for ord_ in (0, 1):
        vals = collections.defaultdict(list)
        for kind in ("f", "t"):
            for f in glob.glob(f"{out}/{kind}_{ord_}_{shape}/*/*counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if "ivit_gemm_bf16_160x128x64" in r["Kernel_Name"]:
                        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        # the timed launches dominate (the few check launches use order 0): medians
        med = {k: sorted(v)[len(v) // 2] for k, v in vals.items() if v}
        fetch = med.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6
        hit = med.get("TCC_HIT_sum", 0); miss = med.get("TCC_MISS_sum", 0)
        print(f"{names[shape]:24s} order {ord_}: HBM-side read {fetch:7.1f} MB per launch, L2 hit rate {hit / max(1.0, hit + miss):.3f}  ({len(vals.get('FETCH_SIZE', []))} launches)")
PY
