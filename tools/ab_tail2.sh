# same-box A/B: tail split rules on config 3 (ViT-L/16-384 B = 128: last rounds 52-55 % full) - 0 off, 1 the shipped rule (no effect here), 2 the wide rule
run() {
  n=$1; shift
  env "$@" > gpurun_out/abt2_$n.json 2> gpurun_out/abt2_$n.err || { tail -5 gpurun_out/abt2_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abt2_$n.json"))
k={}
L=24 if "c3" in "$n" else 32
for x in d["roofline"]["kernels"]:
    r=x["kernel"].split(":")[0]; k[r]=round(k.get(r,0)+x["ms_per_step"]*1e3/ (L if r in ("qkv","proj","mlp1","mlp2","attention") else 1),1)
print("$n", d["value"], d["ms_per_step"], d["parity"]["ok"], d["roofline"]["frac"], {r: k.get(r) for r in ("qkv","proj","mlp1","mlp2")})
PY
}
B="timeout -k 10 400 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 10 --warmup 3"
for i in 1 2; do
  run c3_rule1_$i $B --config 3 || exit 1
  run c3_wide_$i IVIT_GEMM_TAIL=2 $B --config 3 || exit 1
done
run h14_rule1 $B --model vit_h_14 --batch-per-gpu 256 --precision bf16 || exit 1
run h14_wide IVIT_GEMM_TAIL=2 $B --model vit_h_14 --batch-per-gpu 256 --precision bf16 || exit 1
