# config 3 / ViT-H: the LayerNorm fold with one statistics kernel per LayerNorm (the default since round 5) vs rounds 3-4's rule (IVIT_FOLD_LN=3: LayerNorm kernels at these batches)
run() {
  n=$1; shift
  env "$@" > gpurun_out/abl_$n.json 2> gpurun_out/abl_$n.err || { tail -5 gpurun_out/abl_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$n.json"))
k={}
for x in d["roofline"]["kernels"]:
    r=x["kernel"].split(":")[0]; k[r]=round(k.get(r,0)+x["ms_per_step"]*1e3/ ((24 if "c3" in "$n" else 32) if r in ("qkv","proj","mlp1","mlp2","attention") else 1),1)
print("$n", d["value"], d["ms_per_step"], d["parity"]["ok"], d["parity"]["logits_vs_plain_f32_oracle"], {r: k.get(r) for r in ("qkv","proj","mlp1","mlp2","layernorm")})
PY
}
B="timeout -k 10 400 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 10 --warmup 3"
for i in 1 2; do
run c3_base_$i IVIT_FOLD_LN=3 $B --config 3 || exit 1
run c3_fold_$i $B --config 3 || exit 1
done
run h14_base IVIT_FOLD_LN=3 $B --model vit_h_14 --batch-per-gpu 256 --precision bf16 || exit 1
run h14_fold $B --model vit_h_14 --batch-per-gpu 256 --precision bf16 || exit 1
