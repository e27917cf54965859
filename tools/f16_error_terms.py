"""Error budget of the f16 data path on the whole ViT-B/16 chain (VERDICT r2 #1a): the rounding-aware oracle with ONE rounding
point enabled at a time (oracle/vit_oracle.py: ROUND_ONLY), with all of them, and with the remedies (split patch / head GEMMs,
W' rounded once).  CPU only; study tool, not product.   python tools/f16_error_terms.py [model] [images]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images
from oracle import vit_oracle as vo

def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def rms(a, b): return float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())

model = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = torch.float16 if (len(sys.argv) <= 3 or sys.argv[3] == "f16") else torch.bfloat16
cfg = VARIANTS[model]
sd = init_weights(cfg, seed=0, mode="spec")
x = synthetic_images(64, cfg, seed=1234)[:nimg]
torch.set_num_threads(os.cpu_count())
ref = vo.forward(x.double(), sd, cfg)["logits"]
ref32 = vo.forward(x, sd, cfg)["logits"]
print(f"{model}, {nimg} images, operand type {dtype}; plain f32 vs plain f64 logits: max {rel(ref32, ref):.2e}")
vo.OPERAND_DTYPE = dtype
vo.LN_FOLD = True
points = ["patch", "w_patch", "x", "w_qkv", "qkv", "p", "att", "w_proj", "w_mlp1", "gelu", "w_mlp2", "head_in", "w_head"]
def run(only, split=(), fold="twice"):
    vo.ROUND_ONLY, vo.SPLIT_GEMMS, vo.FOLD_ROUNDING = only, frozenset(split), fold
    t0 = time.time()
    out = vo.forward(x.double(), sd, cfg, emulate=True)["logits"]
    vo.ROUND_ONLY, vo.SPLIT_GEMMS, vo.FOLD_ROUNDING = None, frozenset(), "twice"
    return rel(out, ref), rms(out, ref), time.time() - t0
print(f"{'rounding points enabled':58s} max-norm    rms")
tot2 = 0.0
for p in (points if "--terms" in sys.argv else []):
    m, r, dt = run({p}); tot2 += r * r
    print(f"{p:58s} {m:.2e}  {r:.2e}   ({dt:.0f} s)", flush=True)
print(f"{'root sum of squares of the rms column':58s}           {tot2 ** 0.5:.2e}")
for name, kw in [("ALL (the engine as it was in round 2)", dict(only=None)),
                 ("all, patch + head GEMMs split hi/lo", dict(only=None, split=("patch", "head"))),
                 ("all, patch + head + proj GEMMs split hi/lo", dict(only=None, split=("patch", "head", "proj"))),
                 ("all, patch + head + proj + mlp1w", dict(only=None, split=("patch", "head", "proj", "mlp1w"))),
                 ("all, patch + head + proj + mlp2w", dict(only=None, split=("patch", "head", "proj", "mlp2w"))),
                 ("all, patch + head + proj + mlp1w + mlp2w", dict(only=None, split=("patch", "head", "proj", "mlp1w", "mlp2w"))),
                 ("all, patch + head + mlp1w + mlp2w", dict(only=None, split=("patch", "head", "mlp1w", "mlp2w"))),
                 ("all, patch + head + proj + qkvw + mlp1w + mlp2w", dict(only=None, split=("patch", "head", "proj", "qkvw", "mlp1w", "mlp2w"))),
                 ("all, W' = rn(W gamma) rounded once", dict(only=None, fold="once")),
                 ("all but the weights (activations only)", dict(only={"patch", "x", "qkv", "p", "att", "gelu", "head_in"})),
                 ("weights only", dict(only={"w_patch", "w_qkv", "w_proj", "w_mlp1", "w_mlp2", "w_head"}))]:
    m, r, dt = run(**kw)
    print(f"{name:58s} {m:.2e}  {r:.2e}", flush=True)
