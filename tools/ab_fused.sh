# in-forward A/B (bench.py, alternating, one box): the fused MLP kernel against the two GEMM launches, and the f16x split sets through it
run() {
  n=$1; shift
  env "$@" > gpurun_out/r05_fab_$n.json 2> gpurun_out/r05_fab_$n.err || { tail -5 gpurun_out/r05_fab_$n.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_fab_$n.json").read().strip().splitlines()[-1])
k=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("mlp") or x["kernel"].startswith("proj")]
print("$n", d["value"], d["ms_per_step"], d.get("parity",{}).get("ok"), d.get("parity",{}).get("logits_vs_plain_f32_oracle"), [(x["kernel"], x["avg_us"]) for x in k])
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 100"
for i in 1 2; do
run bf16_fused_$i $B || exit 1
run bf16_pair_$i IVIT_FUSED_MLP=0 $B || exit 1
run f16x_default_$i $B --precision f16x || exit 1                       # up weight + out-projection as pairs
run f16x_noproj_$i IVIT_F16X_PROJ=0 $B --precision f16x || exit 1
run f16x_both_$i IVIT_F16X_MLP2=1 $B --precision f16x || exit 1       # round 4's set
done
