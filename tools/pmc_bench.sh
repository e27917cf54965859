#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats            -> per-kernel durations (profiles/<tag>_kernel_stats.csv)
#   2. --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md "HBM" section)
#      -> HBM bytes per launch per kernel (profiles/<tag>_pmc_traffic.json), FETCH_SIZE doubled as the
#         guide prescribes for wide coalesced streams on gfx950 (it tallies 128-B requests at 64 B).
# usage: tools/pmc_bench.sh <tag> [bench args]
set -e
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/write.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
grep "^{\"metric\"" $OUT/stats.log | tail -1 > profiles/${TAG}_bench_under_rocprof.json || true
python3 - "$OUT" "$TAG" "$*" <<'PY'
import csv, glob, json, sys, collections, re
out, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3]
def opt(name, default):
    m = re.search(name + r"[ =](\S+)", extra)
    return m.group(1) if m else default
configs = {"2": ("vit_b_16", "64", "bf16"), "3": ("vit_l_16_384", "128", "bf16"), "4": ("vit_b_16", "256", "bf16"), "5": ("vit_h_14", "256", "fp8")}   # bench.py CONFIGS
cm, cb, cp = configs[opt('--config', '2')]
workload = f"--model {opt('--model', cm)} --batch-per-gpu {opt('--batch-per-gpu', cb)} --precision {opt('--precision', cp)}"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("ivit::"): continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(agg.items()):
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", [])
    fetch_kb = sum(f) / len(f) if f else None       # counter unit: KiB
    write_kb = sum(w) / len(w) if w else None
    res[k] = {"launches_profiled": len(f) or len(w),
              "FETCH_SIZE_KiB_mean_raw": fetch_kb, "WRITE_SIZE_KiB_mean_raw": write_kb,
              "hbm_read_bytes_per_launch": None if fetch_kb is None else 2.0 * fetch_kb * 1024,   # gfx950: x2
              "hbm_write_bytes_per_launch": None if write_kb is None else write_kb * 1024}
    r = res[k]
    if r["hbm_read_bytes_per_launch"] is not None and r["hbm_write_bytes_per_launch"] is not None:
        r["hbm_bytes_per_launch"] = r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --steps 10; FETCH_SIZE x2 (gfx950 correction)",
           "workload": workload, "kernels": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
for k, r in res.items():
    print(f"{k:50s} read {r['hbm_read_bytes_per_launch']}  write {r['hbm_write_bytes_per_launch']}")
PY
