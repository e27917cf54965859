"""The RCCL leg on the GPU box (one GPU available here): bench.py's multi-rank code path with a
world of ONE rank over backend "nccl" (= RCCL on ROCm) - process-group init on the device, the single
all-gather of the packed [logits | class features] block, barrier + max-over-ranks timing.  The
N = 2 logic itself is covered on CPU with gloo (tests/test_distributed.py); the N = 8 run is the
driver's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_runs_under_torchrun_with_rccl():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IVIT_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--batch-per-gpu", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, out_lines          # RCCL's banner etc. must not reach stdout: ONE line, the result
    d = json.loads(out_lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["collective"].startswith("ncclAllGather issued by the engine (ivit_allgather_rows, RCCL)")
    assert d["parity"]["gathered_equals_local"] is True


def test_bench_json_contract_single_gpu():
    """`python bench.py` (N = 1): one JSON line with the contract's keys, the roofline and
    cpu_baseline objects, and the parity gate evaluated in the same run."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "images/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert "vit_b_16" in d["config"]["workload"] and d["config"]["batch_per_gpu"] == 64 and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2516.6
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert abs(d["value"] - 64 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.01
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "images/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["parity"]["logits_vs_plain_f32_oracle"] <= 9e-3                    # ViT-B/16 whole chain: measured 7.3e-3 + 25 %
    for node, rec in d["parity"]["per_node"].items():
        assert rec["vs_rounding_oracle"] <= 1e-3 and rec["vs_plain_f32"] <= 3.5e-3, (node, rec)
    # the additions of round 2: per-kernel roofline list, step-time distribution, PCIe-inclusive rate, device facts
    names = [k["kernel"] for k in rf["kernels"]]
    assert any(n.startswith("mlp:ivit_mlp_fused_bf16") for n in names) and any(n.startswith("attention") for n in names), names
    assert all(0.0 < k["frac"] < 1.0 for k in rf["kernels"] if "frac" in k)
    assert rf["device"]["compute_units"] >= 1
    sm = d["step_ms"]
    assert sm["steps"] >= 50 and sm["p10"] <= sm["median"] <= sm["p90"]
    assert 0 < d["pcie_inclusive"]["value"] < d["value"]
    assert "vit_ti_16_byte_path_requests_per_s" in cb["extra"] and cb["extra"]["vit_b_16_batch1_images_per_s"] > 0


def test_bench_config4_shard_on_one_gpu():
    """BASELINE configs[3] is ViT-B/16 B = 2048 over 8 GPUs = 256 images per GPU: `--config 4` selects that shard size;
    one rank of it runs here (the 8-rank run is the driver's)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["batch_per_gpu"] == 256 and d["config"]["baseline_config"] == 4 and d["n_gpus"] == 1
    assert d["value"] > 0 and d["parity"]["logits_vs_plain_f32_oracle"] <= 9e-3


def test_bench_gpus_n_is_launched_by_bench_itself():
    """`python bench.py --gpus 2` un-wrapped: bench.py starts the ranks itself (a child torch.distributed.run).  This box
    has one GPU, so rank 1 must stop with a clear device error and the parent must report failure - no hang, no
    result line, no re-exec of a process that touched the GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--batch-per-gpu", "2"]
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU box: the real N = 2 run is the driver's")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_packed_forward_writes_the_collective_block_in_place():
    """the multi-GPU step is forward + ONE ncclAllGather.  ivit_forward_device_packed lets the head GEMM and the final
    LayerNorm store [logits | class-token features] with the packed block's row stride: bit-identical to the two dense outputs,
    nothing outside the block's columns touched; ivit_allgather_rows on a communicator of one rank hands the block back unchanged."""
    import torch
    from interactive_vit_amd.engine import Engine
    from interactive_vit_amd.vit_config import test_config as small_config
    from interactive_vit_amd.weights import init_weights, synthetic_images
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=5)
    try:
        b = 5
        x = synthetic_images(b, cfg, seed=17).cuda()
        ns = len(eng.stages)
        stream = torch.cuda.current_stream().cuda_stream
        logits = torch.empty((b, cfg.classes), dtype=torch.float32, device="cuda")
        clsf = torch.empty((b, cfg.dim), dtype=torch.float32, device="cuda")
        eng.forward_into(x, logits, clsf, b, 0, ns, stream)
        width = cfg.classes + cfg.dim
        packed = torch.full((b, width + 8), -7.0, dtype=torch.float32, device="cuda")     # a stride wider than the block
        eng.forward_packed(x, packed, b, 0, stream)
        torch.cuda.synchronize()
        assert torch.equal(packed[:, :cfg.classes], logits) and torch.equal(packed[:, cfg.classes:width], clsf)
        assert bool((packed[:, width:] == -7.0).all())
        tight = torch.empty((b, width), dtype=torch.float32, device="cuda")
        eng.forward_packed(x, tight, b, 0, stream)
        eng.comm_init(0, 1, lambda ident: ident)
        out = torch.zeros((b, width), dtype=torch.float32, device="cuda")
        eng.allgather_rows(tight, b, out, stream)
        torch.cuda.synchronize()
        assert torch.equal(out[:, :cfg.classes], logits) and torch.equal(out[:, cfg.classes:], clsf)
        with pytest.raises(Exception, match="row_stride"):
            eng.forward_packed(x, torch.empty((b, width - 4), dtype=torch.float32, device="cuda"), b, 0, stream)
    finally:
        eng.close()


def test_patch_outputs_gather_through_the_same_collective():
    """north_star: "a single RCCL all-gather ... to reassemble the [CLS]/patch outputs for the interactive view".  ivit_allgather_rows takes any
    row width: the [b, N, D] patch outputs of encoder.ln go through it as rows of N * D floats, on a SIDE stream (issued
    asynchronously beside the compute stream), here on a communicator of one rank - the block comes back unchanged, in image order."""
    import torch
    from interactive_vit_amd.engine import Engine
    from interactive_vit_amd.vit_config import test_config as small_config
    from interactive_vit_amd.weights import init_weights, synthetic_images
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=3)
    try:
        b = 3
        x = synthetic_images(b, cfg, seed=23).cuda()
        ln_stage = eng.stage_index("encoder.ln")
        patches = eng.forward(x, 0, ln_stage + 1)                       # [b, N, D] f32: every token after the final LayerNorm
        assert patches.shape == (b, cfg.tokens, cfg.dim)
        eng.comm_init(0, 1, lambda ident: ident)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        rows = patches.reshape(b, cfg.tokens * cfg.dim).contiguous()
        out = torch.zeros_like(rows)
        eng.allgather_rows(rows, b, out, side.cuda_stream)
        side.synchronize()
        assert torch.equal(out.reshape(b, cfg.tokens, cfg.dim), patches)
    finally:
        eng.close()
