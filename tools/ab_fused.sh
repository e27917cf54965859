# in-forward A/B (bench.py, alternating): fused MLP kernel on / off on the configurations that can take it
run() {
  n=$1; shift
  env "$@" > gpurun_out/r05_fab_$n.json 2> gpurun_out/r05_fab_$n.err || { tail -5 gpurun_out/r05_fab_$n.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_fab_$n.json").read().strip().splitlines()[-1])
k=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("mlp")]
print("$n", d["value"], d["ms_per_step"], d.get("parity",{}).get("ok"), d.get("parity",{}).get("logits_vs_plain_f32_oracle"), [(x["kernel"], x["avg_us"]) for x in k])
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-tolerance-mode"
for i in 1 2; do
run c4_unfused_$i IVIT_FUSED_MLP=0 $B --config 4 --steps 30 || exit 1
run c4_fused_$i IVIT_FUSED_MLP=1 $B --config 4 --steps 30 || exit 1
run f16x_unfused_$i IVIT_FUSED_MLP=0 $B --precision f16x --steps 60 || exit 1
run f16x_fused_$i IVIT_FUSED_MLP=1 $B --precision f16x --steps 60 || exit 1
run f16_unfused_$i IVIT_FUSED_MLP=0 $B --precision f16 --steps 60 || exit 1
run f16_fused_$i IVIT_FUSED_MLP=1 $B --precision f16 --steps 60 || exit 1
done
