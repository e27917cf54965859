# same-box A/B: column-panel width of the 256 x 256 tiles' block -> tile map (IVIT_GEMM_GROUP_N; default 8), ViT-B/16 B = 64 (QKV: 9 column tiles) and config 3
run() {
  n=$1; shift
  env "$@" > gpurun_out/abg_$n.json 2> gpurun_out/abg_$n.err || { tail -5 gpurun_out/abg_$n.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abg_$n.json"))
k={x["kernel"].split(":")[0]: x["avg_us"] for x in d["roofline"]["kernels"]}
print("$n", d["value"], d["ms_per_step"], d["parity"]["ok"], {r: k.get(r) for r in ("qkv","proj","mlp","mlp1","mlp2")})
PY
}
B="timeout -k 10 300 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg"
for i in 1 2; do
  for g in 8 9 5 4; do run b16_g${g}_$i IVIT_GEMM_GROUP_N=$g $B --steps 100 || exit 1; done
done
for i in 1 2; do
  for g in 8 4 6 12 16; do run c3_g${g}_$i IVIT_GEMM_GROUP_N=$g $B --config 3 --steps 10 --warmup 3 || exit 1; done
done
