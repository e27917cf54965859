# same-box A/B: config 4 (ViT-B/16, 256 images per GPU: 788 row blocks = 3 rounds + 20) - the fused MLP kernel's last 20 blocks as the GEMM pair (IVIT_MLP_TAIL, default on)
run() {
  n=$1; shift
  env "$@" > gpurun_out/abm_$n.json 2> gpurun_out/abm_$n.err || { tail -5 gpurun_out/abm_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abm_$n.json"))
k={}
for x in d["roofline"]["kernels"]:
    r=x["kernel"].split(":")[0]; k[r]=round(k.get(r,0)+x["ms_per_step"]*1e3/ (12 if r in ("qkv","proj","mlp","mlp1","mlp2","attention") else 1),1)
print("$n", d["value"], d["ms_per_step"], d["parity"]["ok"], {r: k.get(r) for r in ("mlp","mlp1","mlp2","qkv","proj")})
PY
}
B="timeout -k 10 400 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 30 --config 4"
for i in 1 2; do
  run tail_$i $B || exit 1
  run off_$i IVIT_MLP_TAIL=0 $B || exit 1
done
