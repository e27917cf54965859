"""Practical-ceiling probe: times the vendor GEMM (hipBLASLt via torch.mm / F.linear) on the ViT shapes,
for comparison with tools/gemm_bench.bin.  Measurement aid only; nothing in the product uses it."""
import sys
import torch

shapes = [("qkv", 12608, 2304, 768), ("proj", 12608, 768, 768), ("mlp1", 12608, 3072, 768), ("mlp2", 12608, 768, 3072),
          ("h_qkv", 65792, 3840, 1280), ("h_mlp1", 65792, 5120, 1280), ("h_mlp2", 65792, 1280, 5120)]
dev = "cuda:0"
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for label, fn in (("mm", lambda: torch.mm(a, w.t())), ("linear+bias", lambda: torch.nn.functional.linear(a, w, b))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name:7s} M={M} N={N} K={K} {label:12s} median {med*1e3:8.2f} us  {2.0*M*N*K/med/1e9:7.1f} TF/s", flush=True)
