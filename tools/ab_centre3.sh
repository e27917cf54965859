# same-box A/B after moving the centre loads ahead of the stores: seeded weights plain (default policy) vs centred wherever it lowers the statistic (IVIT_FOLD_CENTRE=2), and realistic weights
run() {
  n=$1; shift
  env "$@" > gpurun_out/abc3_$n.json 2> gpurun_out/abc3_$n.err || { tail -5 gpurun_out/abc3_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abc3_$n.json"))
k={x["kernel"].split(":")[0]: x["avg_us"] for x in d["roofline"]["kernels"]}
print("$n", d["value"], d["ms_per_step"], d["parity"]["logits_vs_plain_f32_oracle"], d["config"]["layernorm"][-70:], {r: k.get(r) for r in ("mlp","qkv","proj")})
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 100"
for i in 1 2 3; do
  run spec_default_$i $B || exit 1
  run spec_centre2_$i IVIT_FOLD_CENTRE=2 $B || exit 1
  run realistic_$i $B --weights realistic || exit 1
done
