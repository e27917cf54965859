import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images
from oracle import vit_oracle as vo
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def rms(a, b): return float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
model = sys.argv[1]; nimg = int(sys.argv[2]); seeds = [int(s) for s in sys.argv[3].split(",")]
cfg = VARIANTS[model]
torch.set_num_threads(int(os.environ.get("NT", "4")))
vo.OPERAND_DTYPE = torch.float16
vo.LN_FOLD = bool(int(os.environ.get("FOLD", "1")))
sets = [("ph", ("patch","head")),
        ("ph+mlp1w", ("patch","head","mlp1w")),
        ("ph+mlp1w+mlp2w", ("patch","head","mlp1w","mlp2w")),
        ("ph+proj+mlp1w+mlp2w", ("patch","head","proj","mlp1w","mlp2w")),
        ("ph+proj+qkvw+mlp1w+mlp2w", ("patch","head","proj","qkvw","mlp1w","mlp2w"))]
for seed in seeds:
    sd = init_weights(cfg, seed=seed, mode="spec")
    x = synthetic_images(nimg, cfg, seed=1234 + seed)
    t0 = time.time()
    ref = vo.forward(x.double(), sd, cfg)["logits"]
    print(f"{model} seed {seed} {nimg} img: f64 forward {time.time()-t0:.0f}s", flush=True)
    for name, sp in sets:
        vo.SPLIT_GEMMS = frozenset(sp)
        out = vo.forward(x.double(), sd, cfg, emulate=True)["logits"]
        per = [rel(out[i:i+1], ref[i:i+1]) for i in range(nimg)]
        print(f"  {name:28s} max {rel(out, ref):.2e} rms {rms(out, ref):.2e} per-image max {max(per):.2e}", flush=True)
    vo.SPLIT_GEMMS = frozenset()
