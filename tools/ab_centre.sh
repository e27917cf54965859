# same-box A/B: the residual epilogue's 16-bit copy beside the f32 stores (default) vs after the statistics (IVIT_RS_COPY_LAST=1 build), spec and realistic weights
B="timeout -k 10 200 python bench.py --no-cpu-baseline --no-tolerance-mode --no-layernorm-leg --steps 100"
for i in 1 2 3; do
  for v in default copylast; do
    lib=interactive_vit_amd/libivit.so; [ $v = copylast ] && lib=tools/libivit_copylast.so
    for w in spec realistic; do
      IVIT_LIB=$lib $B --weights $w > gpurun_out/abc_${v}_${w}_$i.json 2> gpurun_out/abc.err || { tail -3 gpurun_out/abc.err; exit 1; }
      python - <<PY
import json
d=json.load(open("gpurun_out/abc_${v}_${w}_$i.json"))
k={x["kernel"].split(":")[0]: x["avg_us"] for x in d["roofline"]["kernels"]}
print("$v $w $i", d["value"], d["ms_per_step"], {r: k.get(r) for r in ("mlp","qkv","proj","attention")})
PY
    done
  done
done
