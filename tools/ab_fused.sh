run() { # name, env...
  n=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-tolerance-mode > gpurun_out/r05_fab2_$n.json 2> gpurun_out/r05_fab2_$n.err || { tail -5 gpurun_out/r05_fab2_$n.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_fab2_$n.json").read().strip().splitlines()[-1])
k=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("mlp")]
print("$n", d["value"], d["ms_per_step"], d.get("parity",{}).get("ok"), [(x["kernel"].split(":")[0], x["avg_us"]) for x in k])
PY
}
for i in 1 2; do
run unfused_$i IVIT_FUSED_MLP=0 || exit 1
run touch_$i IVIT_FUSED_MLP=1 || exit 1
run notouch_$i IVIT_FUSED_MLP=1 IVIT_LIB=tools/libivit_t0.so || exit 1
done
