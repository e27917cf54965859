#!/bin/bash
# SQ/TCC counters per kernel over a short bench run (separate passes).  usage: tools/pmc_kernels.sh <outdir> [tag]
# with a tag: also writes profiles/<tag>_pmc_mfma.json (MFMA utilisation, wave-cycle split, L2 hit rate per kernel)
OUT=${1:-gpurun_out/pmc_k}; TAG=$2; mkdir -p $OUT profiles; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg"
run() { rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $OUT/$2 -- python3 bench.py $ARGS > $OUT/$2.log 2>&1 || echo "pass $2 failed"; }
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" sq
run "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" sq2
run "TCC_HIT_sum TCC_MISS_sum" tcc
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
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
if tag:
    res = {}
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        r = {"launches_profiled": max(len(v) for v in d.values()), "counters_mean_per_launch": m}
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            # SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD summed
            # over SIMDs (MI355X_MICROARCH.md, cycle constants): MFMA utilisation = busy cycles / (SQ_BUSY_CYCLES-based wall x SIMDs) is not
            # derivable without the per-SE layout, so the ratios below are the portable ones
            r["wait_any_frac_of_wave_cycles"] = m.get("SQ_WAIT_ANY", 0.0) / wc
            r["wait_inst_any_frac_of_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0.0) / wc
            r["active_inst_any_frac_of_wave_cycles"] = m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; 1024 SIMDs share that wall time
            r["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) > 0:
            r["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        res[k] = r
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ / TCC counters>, three separate passes of bench.py --steps 3 (tools/pmc_kernels.sh)",
               "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): fraction of SIMD-cycles with the matrix pipe busy; wait_any = waves parked in s_waitcnt / s_barrier",
               "kernels": res}, open(f"profiles/{tag}_pmc_mfma.json", "w"), indent=1)
PY
