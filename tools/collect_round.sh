#!/bin/bash
# One round's measurement set (run on the GPU box from the repo root): usage tools/collect_round.sh <tag>
#   default bench (+ cpu baseline), rocprofv3 kernel stats + PMC traffic + MFMA counters of the default workload, and the other BASELINE configurations.
TAG=${1:-r05b}; mkdir -p gpurun_out/$TAG profiles
B="timeout -k 10 500 python bench.py"
$B > gpurun_out/$TAG/bench_vit_b_16.json 2> gpurun_out/$TAG/bench_vit_b_16.err || echo "default bench failed"
timeout -k 10 700 bash tools/pmc_bench.sh $TAG > gpurun_out/$TAG/pmc.log 2>&1 || echo "pmc_bench failed"
timeout -k 10 400 bash tools/pmc_kernels.sh gpurun_out/$TAG/pmc_k $TAG > gpurun_out/$TAG/pmc_k.log 2>&1 || echo "pmc_kernels failed"
for spec in "c3:--config 3" "c4:--config 4" "c5:--config 5" "h14_bf16:--model vit_h_14 --batch-per-gpu 256 --precision bf16" "f16:--precision f16" "f16x:--precision f16x" "realistic:--weights realistic"; do
  n=${spec%%:*}; a=${spec#*:}
  $B --no-cpu-baseline --steps 20 $a > gpurun_out/$TAG/bench_$n.json 2> gpurun_out/$TAG/bench_$n.err || echo "bench $n failed"
  echo "done $n"
done
cp profiles/${TAG}_* gpurun_out/$TAG/ 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$TAG/bench_*.json")):
    try:
        d = json.load(open(f))
        t = d.get("tolerance_mode") or {}
        l = d.get("layernorm_kernels") or {}
        print(f.split("/")[-1], d["value"], d["roofline"]["frac"], d["parity"]["ok"], t.get("value"), t.get("logits_vs_plain_f32_oracle"), l.get("value"))
    except Exception as ex:
        print(f, "unreadable", ex)
PY
