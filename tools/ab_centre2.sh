# same-box A/B: what the centred copies cost on the SEEDED weights (where the calibration centres the sites it helps): default vs IVIT_FOLD_CENTRE=0, bf16 and f16x
run() {
  n=$1; shift
  env "$@" > gpurun_out/abc2_$n.json 2> gpurun_out/abc2_$n.err || { tail -5 gpurun_out/abc2_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abc2_$n.json"))
k={x["kernel"].split(":")[0]: x["avg_us"] for x in d["roofline"]["kernels"]}
print("$n", d["value"], d["ms_per_step"], d["parity"]["logits_vs_plain_f32_oracle"], d["config"]["layernorm"][-60:], {r: k.get(r) for r in ("mlp","qkv","proj")})
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 100"
for i in 1 2 3; do
  run bf16_centre_$i $B || exit 1
  run bf16_plain_$i IVIT_FOLD_CENTRE=0 $B || exit 1
done
for i in 1 2; do
  run f16x_centre_$i $B --precision f16x || exit 1
  run f16x_plain_$i IVIT_FOLD_CENTRE=0 $B --precision f16x || exit 1
done
