# same-box A/B: rows of a nearly empty last round of 256 x 256 tiles peeled into their own launch (IVIT_GEMM_TAIL, default on) - ViT-H/14 B = 256 bf16 and fp8
run() {
  n=$1; shift
  env "$@" > gpurun_out/abt_$n.json 2> gpurun_out/abt_$n.err || { tail -5 gpurun_out/abt_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abt_$n.json"))
k={}
for x in d["roofline"]["kernels"]:
    r=x["kernel"].split(":")[0]; k[r]=round(k.get(r,0)+x["ms_per_step"]*1e3/ (32 if r in ("qkv","proj","mlp1","mlp2","attention") else 1),1)
print("$n", d["value"], d["ms_per_step"], d["parity"]["ok"], d["roofline"]["frac"], {r: k.get(r) for r in ("qkv","proj","mlp1","mlp2","attention")})
PY
}
B="timeout -k 10 400 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 10 --warmup 3"
for i in 1 2; do


  run c5_fp8_tail_$i $B --config 5 || exit 1
  run c5_fp8_off_$i IVIT_GEMM_TAIL_FP8=0 $B --config 5 || exit 1
done
