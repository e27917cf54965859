"""What bounds the fp8 (e4m3 W + A) data path's distance from the f32 forward on ViT-H/14: the oracle's fp8 forward
with the engine's scaling policy (per-tensor static activation scales, per-row weight scales) against MX-style per-32-element e8m0
block scales for the activations, per-token scales, and with only the weights or only the activations quantised.  CPU only.
   python tools/fp8_error_terms.py [model] [images] [layers]"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images
from oracle import vit_oracle as vo

model = sys.argv[1] if len(sys.argv) > 1 else "vit_h_14"
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = VARIANTS[model]
if len(sys.argv) > 3:
    import dataclasses
    cfg = dataclasses.replace(cfg, layers=int(sys.argv[3]))
sd = init_weights(cfg, seed=0, mode="spec")
x = synthetic_images(nimg, cfg, seed=1234)
torch.set_num_threads(os.cpu_count())
def rel(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
def rms(a, b): return float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())

def e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype)

def q_tensor(t, scale):           # the engine's policy: one static scale per tensor
    return e4m3(t / scale) * scale
def q_row(t):                     # per token row
    s = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    return e4m3(t / s) * s
def q_mx(t, block=32):            # e8m0 (power of two) scale per `block` consecutive elements of a row
    shp = t.shape
    tb = t.reshape(*shp[:-1], shp[-1] // block, block)
    amax = tb.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -120)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    return (e4m3(tb / s) * s).reshape(shp)
def q_none(t): return t

def w_q(sd, key, dt, mode):
    if mode == "none":
        return vo._w(sd, key, dt, True)       # bf16 copy only
    return vo.q8_weight(sd, key, dt)

def forward(actq, wmode, scales=None):
    dt = torch.float64
    t = vo.tokens(vo.conv_proj(vo.transform(x.to(torch.float32)).to(dt), sd, cfg, True), sd, cfg)
    for i in range(cfg.layers):
        pre = vo.layer_prefix(i)
        b, n, d = t.shape
        sc = scales[4 * i:4 * i + 4] if scales is not None else [None] * 4
        qa = (lambda v, k: q_tensor(v, sc[k])) if actq == "tensor" else (lambda v, k: {"row": q_row, "mx": q_mx, "none": q_none}[actq](v))
        h = qa(vo.layer_norm(t, vo._w(sd, pre + "ln_1.weight", dt), vo._w(sd, pre + "ln_1.bias", dt), cfg.ln_eps), 0)
        qkv = vo.rnd(h @ w_q(sd, pre + "self_attention.in_proj_weight", dt, wmode).t() + vo._w(sd, pre + "self_attention.in_proj_bias", dt), True, torch.bfloat16)
        a, _ = vo.attention_core(qkv, cfg, emulate=True, p_dtype=torch.bfloat16)
        a = qa(a, 1)
        t = t + a @ w_q(sd, pre + "self_attention.out_proj.weight", dt, wmode).t() + vo._w(sd, pre + "self_attention.out_proj.bias", dt)
        h = qa(vo.layer_norm(t, vo._w(sd, pre + "ln_2.weight", dt), vo._w(sd, pre + "ln_2.bias", dt), cfg.ln_eps), 2)
        u = qa(vo.gelu_erf(h @ w_q(sd, pre + "mlp.0.weight", dt, wmode).t() + vo._w(sd, pre + "mlp.0.bias", dt)), 3)
        t = t + u @ w_q(sd, pre + "mlp.3.weight", dt, wmode).t() + vo._w(sd, pre + "mlp.3.bias", dt)
    return vo.heads(vo.cls(vo.encoder_ln(t, sd, cfg)), sd, True)

t0 = time.time()
ref = vo.forward(x, sd, cfg)["logits"]
bf = vo.forward(x.double(), sd, cfg, emulate=True)["logits"]
print(f"{cfg.name} ({cfg.layers} layers), {nimg} images; plain f32 forward {time.time() - t0:.0f} s; bf16 data path vs f32: max {rel(bf, ref):.2e} rms {rms(bf, ref):.2e}")
scales = vo.fp8_calibration_scales(x.double(), sd, cfg)
rows = [("engine policy: e4m3 W (per row) + A (per tensor, static)", "tensor", "q"),
        ("e4m3 W (per row) + A per token row (dynamic)", "row", "q"),
        ("e4m3 W (per row) + A with e8m0 scales per 32 elements (MX)", "mx", "q"),
        ("e4m3 A only (MX), weights bf16", "mx", "none"),
        ("e4m3 W only (per row), activations bf16-free", "none", "q")]
print(f"{'policy':66s} max-norm    rms")
for name, actq, wmode in rows:
    t0 = time.time()
    out = forward(actq, wmode, scales)
    print(f"{name:66s} {rel(out, ref):.2e}  {rms(out, ref):.2e}   ({time.time() - t0:.0f} s)", flush=True)
