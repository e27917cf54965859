"""Probe (round 3): does running the batch as L independent lanes on L HIP streams (each lane B / L images through its own
workspaces) beat one lane of B images?  The GEMM launches are phased (prologue - loop - epilogue, all workgroups in step);
lanes on different streams put one launch's epilogue under another's loop.  Prints img/s for lanes = 1, 2, 4.
    python tools/two_stream_probe.py [--batch 64] [--steps 100]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from interactive_vit_amd.engine import Engine
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_b_16")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--precision", default="bf16")
args = ap.parse_args()
cfg = VARIANTS[args.model]
sd = init_weights(cfg, seed=0, mode="spec")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for rep in range(2):
    for lanes in (1, 2, 4):
        Bl = args.batch // lanes
        engs = [Engine(cfg, sd, device=0, max_batch=Bl, precision=args.precision) for _ in range(lanes)]
        ns = len(engs[0].stages)
        xs = [synthetic_images(Bl, cfg, seed=1234 + i, device="cuda:0") for i in range(lanes)]
        lg = [torch.empty((Bl, cfg.classes), dtype=torch.float32, device=dev) for _ in range(lanes)]
        cl = [torch.empty((Bl, cfg.dim), dtype=torch.float32, device=dev) for _ in range(lanes)]
        st = [torch.cuda.Stream(dev) for _ in range(lanes)]

        def step():
            for i in range(lanes):
                engs[i].forward_into(xs[i], lg[i], cl[i], Bl, 0, ns, st[i].cuda_stream)
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        print(f"lanes={lanes} x {Bl} images: {Bl * lanes * args.steps / el:9.1f} img/s  {el * 1e3 / args.steps:.4f} ms/step", flush=True)
        for e in engs:
            e.close()
        del engs
