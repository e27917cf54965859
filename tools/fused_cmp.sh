set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -x -q -k "fused_qkv" > gpurun_out/fused_test.log 2>&1 || { tail -40 gpurun_out/fused_test.log; exit 1; }
tail -3 gpurun_out/fused_test.log
IVIT_FUSE_QKV=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_unfused.json 2> gpurun_out/bench_unfused.err
IVIT_FUSE_QKV=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
python - <<'PY'
import json
for n in ("unfused","fused"):
    j=json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, j["value"], j["ms_per_step"], j["roofline"]["achieved"])
    for k in j["roofline"]["kernels"]:
        print("   ", k)
PY
