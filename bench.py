#!/usr/bin/env python3
"""Benchmark of the hot path: ViT forward images/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run,
                                                          one rank per GPU, RCCL over xGMI)

One "step" = one forward of one batch of synthetic images through the engine's fused stage range
(transform -> ... -> heads), inputs already resident in HBM, plus - for N > 1 - the single
all-gather that reassembles logits + class-token features on every rank.  At N = 1 the workload is
BASELINE.json configs[1]: ViT-B/16 224^2, bf16, batch 64.  For N > 1 every rank keeps that same
per-GPU batch (weak scaling): images shard by batch, weights are replicated, no other collective.

Rank 0 prints ONE JSON line (contract in the task statement) extended with
  "roofline"     - the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs per launch / average
                   launch duration from HIP events on the launch stream (an instrumented pass of the
                   same K steps, so the headline number is not perturbed), against the dense bf16
                   MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PFLOP/s);
  "cpu_baseline" - the CPU node-graph forward (oracle port: pure torch f32, node by node through
                   Context.compute - the structure the reference executes, main/context.py:143-147)
                   timed on this box's host cores on a bounded sample;
  "parity"       - max|gpu - ref| / max|ref| of the logits against the oracle in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6   # 256 CU x 2.4 GHz x 4096 FLOP/clk/CU, dense
PEAK_FP8_TFLOPS = 5033.2    # dense fp8 (block-scaled MX rate; the non-scaled fp8 MFMA used here issues at the bf16 rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="vit_b_16")
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(cfg, sd, seconds: float):
    """CPU node-graph forward: every node through Context.compute with the oracle backend."""
    from interactive_vit_amd.context import Context, Model, ModelNode
    from interactive_vit_amd.graph import Graph, Pinout
    from interactive_vit_amd.models.vit import make_vit_model_class
    from interactive_vit_amd.weights import synthetic_images
    from oracle.cpu_backend import OracleBackend

    # the box's CPU share, not the host's core count: a 1-GPU box gets 16 cores (oversubscribing
    # 256 threads made one forward take 70 s)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("IVIT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    vit = make_vit_model_class(Model, Pinout)(cfg, OracleBackend(cfg, sd))
    ctx = Context()
    for n in vit.list_node_names():
        ModelNode(vit, n).register(ctx)
    batch = 8
    x = synthetic_images(batch, cfg, seed=1234)

    def one():
        g = Graph()
        nodes = [g.add_node(n, {}) for n in vit.chain_node_names()]
        for a, b in zip(nodes, nodes[1:]):
            g.connect(a, "o", b, "o")
        g.add_input(x, nodes[0], "o")
        ctx.compute(g)
        return nodes[-1].get_pinout().get("o")

    one()  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        one()
        iters += 1
        el = time.perf_counter() - t0
        if el >= seconds or iters >= 200:
            break
    return {"value": batch * iters / el, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{cfg.name} batch {batch} x {iters} forwards, node by node through Context.compute, f32, {el:.1f} s"}


def main():
    args = parse()
    # stdout carries exactly ONE line (the result): everything else that the libraries underneath print
    # there (RCCL's version banner, libdrm notices) is sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} ...`", file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the ViT engine has no CPU path", file=sys.stderr)
        sys.exit(2)

    import torch.distributed as dist
    from interactive_vit_amd.engine import Engine
    from interactive_vit_amd.sharding import all_gather_outputs, shard_range
    from interactive_vit_amd.vit_config import VARIANTS
    from interactive_vit_amd.weights import init_weights, synthetic_images

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # IVIT_FORCE_DIST=1 runs the collective leg even in a world of one rank (RCCL smoke on a 1-GPU box)
    use_dist = world > 1 or (os.environ.get("IVIT_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = VARIANTS[args.model]
    B = args.batch_per_gpu
    total = B * world
    sd = init_weights(cfg, seed=0, mode="spec")                 # replicated weights, seed 0
    eng = Engine(cfg, sd, device=local_rank, max_batch=B, precision=args.precision)
    peak = PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS
    b0, b1 = shard_range(total, rank, world)
    assert b1 - b0 == B
    x = synthetic_images(B, cfg, seed=1234 + rank, device=f"cuda:{local_rank}")   # generated on device
    if args.precision == "fp8":
        eng.calibrate_fp8(x)        # static activation scales + weight quantisation, outside the timed region
    ns = len(eng.stages)
    width = cfg.classes + cfg.dim
    # [logits | cls features] of this rank and the gathered block of the whole batch, double buffered so that
    # the all-gather of step k MAY run on RCCL's stream while step k+1 computes (IVIT_GATHER_OVERLAP=1): a
    # buffer is reused two steps later, after its collective has been waited for, and everything is drained
    # inside the timed region
    packed = [torch.empty((B, width), dtype=torch.float32, device=dev) for _ in range(2)]
    logits = torch.empty((B, cfg.classes), dtype=torch.float32, device=dev)
    clsf = torch.empty((B, cfg.dim), dtype=torch.float32, device=dev)
    gathered = [torch.empty((total, width), dtype=torch.float32, device=dev) for _ in range(2)] if use_dist else None
    pending = [None, None]
    # IVIT_GATHER_OVERLAP=1: leave the collective in flight behind the next step's compute.  Measured on one
    # rank (IVIT_FORCE_DIST=1, same box): no collective 19 166 img/s, in-line collective 18 856-18 941, overlapped
    # 18 605-18 702 - RCCL's kernel then shares the CUs with the forward - so in-line is the default.
    overlap = os.environ.get("IVIT_GATHER_OVERLAP", "0") == "1"
    stream = torch.cuda.current_stream(dev)
    step_no = [0]

    def step():
        eng.forward_into(x, logits, clsf, B, 0, ns, stream.cuda_stream)
        if use_dist:
            k = step_no[0] & 1
            step_no[0] += 1
            if pending[k] is not None:
                pending[k].wait()                               # the collective that last used this buffer pair
            packed[k][:, :cfg.classes].copy_(logits)
            packed[k][:, cfg.classes:].copy_(clsf)
            if overlap:
                pending[k] = all_gather_outputs(packed[k], total, out=gathered[k], async_op=True)   # the ONE collective of the path
            else:
                all_gather_outputs(packed[k], total, out=gathered[k])

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def sync_all():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    drain()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / args.steps
    value = total * args.steps / elapsed
    flops_img = cfg.flops_per_image()

    # ---- instrumented pass: HIP events around every launch, per kernel class (rank 0 only at N=1)
    roofline = None
    classes = None
    if rank == 0:
        eng.profile(True)
        eng.profile_reset()
        for _ in range(args.steps):
            eng.forward_into(x, logits, clsf, B, 0, ns, stream.cuda_stream)
        classes = eng.profile_read()
        eng.profile(False)
        g = classes["gemm"]
        avg_us = g["ms"] * 1e3 / max(1, g["launches"])
        achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        # HBM bytes per GEMM launch from the PMC passes of tools/pmc_bench.sh (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md);
        # bench.py cannot drive the profiler from inside, so it carries the committed summary or null
        traffic = None
        try:
            prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))[-1]
            pj = json.load(open(os.path.join(ROOT, "profiles", prof)))
            ks = pj["kernels"]
            # the counters belong to one workload (summaries without a "workload" key: the default one)
            if pj.get("workload", "--model vit_b_16 --batch-per-gpu 64 --precision bf16") != \
                    f"--model {args.model} --batch-per-gpu {B} --precision {args.precision}":
                ks = {}
            tot = n = 0.0
            for name, rec in ks.items():
                if "gemm" in name and rec.get("hbm_bytes_per_launch"):
                    tot += rec["hbm_bytes_per_launch"] * rec["launches_profiled"]
                    n += rec["launches_profiled"]
            if n:
                traffic = {"hbm_bytes_per_launch": round(tot / n), "source": f"profiles/{prof}"}
        except Exception:
            traffic = None
        roofline = {"bound": "mfma",
                    "kernel": ("ivit_gemm_fp8_{160x128,256x256_stag}x128" if args.precision == "fp8" else
                               "ivit_gemm_bf16_{160x128,256x256_stag}x64[_rs|_lf]") + " (all GEMM launches of the step; tile picked per shape"
                              + ("; _lf / _rs = the LayerNorm-fold epilogues, which carry the LayerNorm work" if (args.precision == "bf16" and eng.ln_fold_for(B)) else "") + ")",
                    "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "launches_per_step": g["launches"] // args.steps, "avg_launch_us": round(avg_us, 2),
                    "algorithmic_gflop_per_step": round(g["flops"] / args.steps / 1e9, 2),
                    "measured": "HIP events on the launch stream around every launch, separate instrumented pass of the same K steps",
                    "e2e_frac": round(value / world * flops_img / 1e12 / peak, 4),
                    "per_class_ms_per_step": {k: round(v["ms"] / args.steps, 4) for k, v in classes.items()}}

    # ---- parity gate in the same run (2 images, oracle on the host)
    parity = None
    cpu = None
    if rank == 0:
        from oracle import vit_oracle as vo
        vo.LN_FOLD = eng.ln_fold_for(B)  # the rounding-aware oracle mirrors the rounding points of THIS batch size
        xs = x[:2].cpu()
        got = logits[:2].cpu().double()
        ref = vo.forward(xs, sd, cfg)["logits"].double()
        if args.precision == "fp8":
            emu = vo.forward_fp8(xs.double(), sd, cfg, eng.fp8_scales())["logits"]
            parity = {"logits_vs_fp8_oracle": float((got - emu).abs().max() / emu.abs().max()),
                      "logits_vs_plain_f32_oracle": float((got - ref).abs().max() / ref.abs().max()),
                      "tolerance_per_node_fp8": 1e-2, "bound_whole_forward_fp8": 1.5e-1, "images": 2}
        else:
            emu = vo.forward(xs.double(), sd, cfg, emulate=True)["logits"]
            parity = {"logits_vs_bf16_rounding_oracle": float((got - emu).abs().max() / emu.abs().max()),
                      "logits_vs_plain_f32_oracle": float((got - ref).abs().max() / ref.abs().max()),
                      "tolerance_per_node": 1e-3, "bound_whole_forward_bf16": 2e-2, "images": 2}
        if use_dist:   # rank 0's shard of the gathered block is exactly what it computed locally
            last = gathered[(step_no[0] - 1) & 1]        # the block of the last step that ran
            parity["gathered_equals_local"] = bool(torch.equal(last[b0:b1, :cfg.classes], logits)
                                                   and torch.equal(last[b0:b1, cfg.classes:], clsf))
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, sd, args.cpu_seconds)

    if use_dist:
        dist.barrier()
    if rank == 0:
        line = {
            "metric": "images/sec ViT-B/16 224^2 forward" if args.model == "vit_b_16" else f"images/sec {args.model} forward",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{cfg.name} {cfg.image}x{cfg.image} forward, batch {B} per GPU (global {total}), "
                                   "f32 images resident in HBM -> f32 logits + class-token features"
                                   + (", one RCCL all-gather per step" if use_dist else ""),
                       "collective": "all_gather_into_tensor over nccl (RCCL), 1 per step" if use_dist else None,
                       "collective_overlap": ("async on RCCL's stream behind the next step's compute, drained inside the timed region" if overlap else "in line") if use_dist else None,
                       "batch_per_gpu": B, "global_batch": total, "tokens": cfg.tokens,
                       "gflop_per_image": round(flops_img / 1e9, 3), "parallelism": f"dp{world}",
                       "weights": "random init N(0,0.02^2) seed 0",
                       "layernorm": "folded into the consuming GEMMs" if (args.precision == "bf16" and eng.ln_fold_for(B)) else "kernel"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    eng.close()
    if use_dist:
        dist.barrier()     # rank 0's parity / instrumented pass runs after the timed loop: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
