"""ORACLE - test infrastructure, not product code.

CPU restatement (pure torch, float32 or float64) of the ViT arithmetic the MI355X engine
implements, node by node.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the product path never does and fails loudly without its HIP library.

What it follows
---------------
The reference (0Marble/interactive-vit) contains no ViT: its server math is "whatever leaf
nn.Module the plugin wraps", called at main/context.py:79-88 (Model.compute -> sub(x)), with
synthetic nodes handled in the plugin's own compute the way static/models/vgg16.py:37-50 does.
The arithmetic therefore lives in the third-party dependency torch (reference pins
pytorch 2.7.0 / torch 2.9.1, requirements.txt:78,89; this container has torch 2.10.0) and the
ViT contract is SURVEY.md Appendix B / section 8(a2'), torchvision ``VisionTransformer`` semantics:

    P[n,k]   = x[c, gy*p+ky, gx*p+kx]      n = gy*G+gx, k = c*p*p + ky*p + kx     (integer-exact)
    t[1+n,:] = P[n,:] @ Wp[D,K]^T + bp ;  t[0,:] = cls ;  t += pos[N,D]
    per layer:  h = LN(t; eps=1e-6); q,k,v = split(h @ Win^T + bin); a = softmax(q k^T/sqrt(dh)) v
                t = t + a @ Wo^T + bo ;  h = LN(t) ;  t = t + GELU_erf(h @ W1^T + b1) @ W2^T + b2
    y = LN(t)[0,:] @ Wh^T + bh

TWO MODES.  ``emulate=False`` is the plain float32 (or float64) forward - "the reference CPU
node-graph forward".  ``emulate=True`` evaluates the SAME formulas but rounds to bfloat16 at exactly
the points where the engine's data path is bf16 (GEMM operands = LN output / unfold image /
attention output / GELU output, weight matrices, the stored q|k|v, the softmax numerators P fed to
P.V); accumulation, LN/softmax statistics, biases and the residual stream stay float32/64 as in the
engine.  bf16 operand rounding alone is a ~1.6e-3 rms relative perturbation of every random-sign
dot product (two operands x 2^-9/sqrt(3)), i.e. ABOVE north_star's 1e-3: against the plain-f32
forward a bf16-MFMA engine cannot be inside 1e-3, against the rounding-aware forward it must be
(only accumulation order differs) - so kernel parity is gated on ``emulate=True`` and the distance
to the plain forward is measured and reported beside it (tests/test_gpu_parity.py).

PARITY PINNING: the reference has no tests and no golden vectors for this path (main/tests.py:1-3
is the empty stub), so the *arithmetic* is pinned by (1) an independent cross-check in
tests/test_oracle.py against torch's own nn.Conv2d / nn.MultiheadAttention / nn.LayerNorm / nn.GELU
modules (the modules a torchvision ViT is assembled from) and (2) golden activations in
tests/golden/ produced by driving this oracle through the REFERENCE's real Request.decode ->
Context.compute -> Response.encode (tests/golden/make_golden.py).  The node plumbing around it
(ordering, wire bytes, Model node naming) is pinned by fixtures generated from the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def layer_prefix(i: int) -> str:
    return f"encoder.layers.encoder_layer_{i}."


# ----------------------------------------------------------------------------------------------
# integer bookkeeping (bit-exact class)
# ----------------------------------------------------------------------------------------------
def unfold_index(image: int, patch: int) -> np.ndarray:
    """int64 [Np, K]: flat offset into a [3,S,S] image of patch n, unfold column k.

    Pure-Python loops on purpose (small, obviously right): n = gy*G+gx, k = c*p*p + ky*p + kx.
    """
    g = image // patch
    k_len = 3 * patch * patch
    idx = np.empty((g * g, k_len), dtype=np.int64)
    for gy in range(g):
        for gx in range(g):
            n = gy * g + gx
            for c in range(3):
                for ky in range(patch):
                    for kx in range(patch):
                        k = c * patch * patch + ky * patch + kx
                        idx[n, k] = c * image * image + (gy * patch + ky) * image + (gx * patch + kx)
    return idx


def unfold(x: torch.Tensor, image: int, patch: int) -> torch.Tensor:
    """[B,3,S,S] -> [B,Np,K] by gathering through ``unfold_index`` (no arithmetic on values)."""
    idx = torch.from_numpy(unfold_index(image, patch)).reshape(-1)
    b = x.shape[0]
    flat = x.reshape(b, -1)
    g = image // patch
    return flat[:, idx].reshape(b, g * g, 3 * patch * patch)


def token_row(batch_index: int, token: int, tokens: int) -> int:
    """Row of (image b, token t) in the engine's [B*N, D] activation matrices."""
    return batch_index * tokens + token


# ----------------------------------------------------------------------------------------------
# floating-point nodes
# ----------------------------------------------------------------------------------------------
# The 16-bit type of the engine's GEMM operand data path that the rounding-aware mode mirrors: bfloat16
# (IVIT_PRECISION_BF16, the default and the BASELINE headline) or float16 (IVIT_PRECISION_F16: same MFMA rate on
# gfx950, 11 significant bits instead of 8 - the mode that meets north_star's 1e-3 against the PLAIN f32 forward).
# Tests / bench set it from Engine.operand_dtype; the plain (emulate=False) forward never looks at it.
OPERAND_DTYPE = torch.bfloat16

# Error-budget studies (tools/f16_error_terms.py): when not None, the rounding-aware mode rounds ONLY at the named
# points - "patch" (unfold image), "w_patch", "x" (the 16-bit copy of the LayerNorm input that the folded GEMMs
# multiply; the LayerNorm output when unfolded), "w_qkv", "w_mlp1" (W' = W . diag(gamma) of the folded GEMMs), "qkv",
# "p" (softmax numerators), "att", "w_proj", "gelu", "w_mlp2", "head_in", "w_head".  None (the default, what every
# test uses) = all of them.
ROUND_ONLY = None

# The split-operand GEMMs of the f16 data path (include/ivit.h: IVIT_PRECISION_F16 / F16X): these GEMMs multiply hi + lo pairs of
# f16 values in two or three MFMA passes with f32 accumulation (x = hi + lo to 22 bits while lo is a normal f16; for |x| < 2^-3
# lo is subnormal with step 2^-24, i.e. ~19 bits for N(0, 0.02^2)-scale weights).  A set of names out of {"patch", "head", "proj"}
# (both operands) and {"qkvw", "mlp1w", "mlp2w"} (weights only); tests / bench set it from Engine.split_gemms.
SPLIT_GEMMS = frozenset()


def _split(name) -> bool:
    return name in SPLIT_GEMMS and OPERAND_DTYPE == torch.float16


def _on(point) -> bool:
    return ROUND_ONLY is None or point is None or point in ROUND_ONLY


def rnd(t: torch.Tensor, emulate: bool, dtype=None, point=None) -> torch.Tensor:
    """Round-to-nearest-even of an activation to the engine's 16-bit operand type at an engine rounding point
    (emulate mode); ``dtype`` pins the type where the engine's own policy does (bf16 q|k|v on the fp8 path)."""
    return t.to(torch.float32).to(dtype or OPERAND_DTYPE).to(t.dtype) if (emulate and _on(point)) else t


def split16(t: torch.Tensor) -> torch.Tensor:
    """hi + lo of the engine's split-operand GEMMs: hi = rn16(t), lo = rn16(t - hi), both in the operand type."""
    t32 = t.to(torch.float32)
    hi = t32.to(OPERAND_DTYPE).to(torch.float32)
    lo = (t32 - hi).to(OPERAND_DTYPE).to(torch.float32)
    return (hi.to(torch.float64) + lo.to(torch.float64)).to(t.dtype)


def _w(sd: Dict[str, torch.Tensor], key: str, dtype, emulate: bool = False, point=None) -> torch.Tensor:
    """A parameter in the compute dtype; ``emulate`` rounds it to the operand type first (weight MATRICES only)."""
    w = sd[key]
    if emulate and _on(point):
        w = w.to(torch.float32).to(OPERAND_DTYPE)
    return w.to(dtype)


def _w16(sd, key: str, dtype, emulate: bool, point, wsplit: bool) -> torch.Tensor:
    """A weight matrix as the engine multiplies it: rounded once (``_w``), or hi + lo of a weight-split GEMM."""
    if emulate and wsplit and _on(point):
        return split16(sd[key]).to(dtype)
    return _w(sd, key, dtype, emulate, point)


def transform(x: torch.Tensor) -> torch.Tensor:
    """Per-channel (x - mean) / std; x is [B,3,S,S] in [0,1]."""
    mean = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def preprocess_resize(cfg) -> int:
    """Shorter-side target of the preset: 256 for a 224 crop, scaled with the model's image size."""
    return (cfg.image * 256 + 112) // 224


def preprocess(x: torch.Tensor, cfg) -> torch.Tensor:
    """The classification preset the reference's model plugin applies to a raw image
    (static/models/vgg16.py:40-42, ``weights.transforms()`` = torchvision ImageClassification):
    [B,3,H,W] in [0,1], any size -> antialiased bilinear resize of the shorter side, centre crop
    S x S, normalise.  Resize arithmetic is ATen's (torch.nn.functional.interpolate, antialias=True);
    sizes and crop offsets follow torchvision's functional resize / center_crop."""
    b, c, h, w = x.shape
    r = preprocess_resize(cfg)
    if h <= w:
        nh, nw = r, int(r * w / h)
    else:
        nh, nw = int(r * h / w), r
    y = torch.nn.functional.interpolate(x.to(torch.float32), size=(nh, nw), mode="bilinear", antialias=True, align_corners=False)
    s_ = cfg.image
    top, left = int(round((nh - s_) / 2.0)), int(round((nw - s_) / 2.0))
    return transform(y[:, :, top:top + s_, left:left + s_]).to(x.dtype)


def conv_proj(x: torch.Tensor, sd, cfg, emulate: bool = False) -> torch.Tensor:
    """Patch embedding as unfold + GEMM: [B,3,S,S] -> [B,Np,D]."""
    dt = x.dtype
    if emulate and _split("patch"):   # three-pass hi/lo product: (hi + lo) of both operands, the lo.lo term dropped
        p = unfold(x, cfg.image, cfg.patch)
        w = sd["conv_proj.weight"].to(dt).reshape(cfg.dim, -1)
        return _split_product(p, w) + _w(sd, "conv_proj.bias", dt)
    p = rnd(unfold(x, cfg.image, cfg.patch), emulate, point="patch")
    w = _w(sd, "conv_proj.weight", dt, emulate, "w_patch").reshape(cfg.dim, -1)
    return p @ w.t() + _w(sd, "conv_proj.bias", dt)


def _split_product(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a @ w.T as the engine's three-pass split GEMM evaluates it: a = ah + al, w = wh + wl in the 16-bit operand
    type; ah.wh + ah.wl + al.wh (the al.wl term, 2^-22 relative, is not computed)."""
    def hl(t):
        t32 = t.to(torch.float32)
        hi = t32.to(OPERAND_DTYPE).to(torch.float32)
        lo = (t32 - hi).to(OPERAND_DTYPE).to(torch.float32)
        return hi.to(t.dtype), lo.to(t.dtype)
    ah, al = hl(a)
    wh, wl = hl(w)
    return ah @ wh.t() + (ah @ wl.t() + al @ wh.t())


def tokens(t: torch.Tensor, sd, cfg) -> torch.Tensor:
    """Prepend the class token, add the learned position embedding: [B,Np,D] -> [B,N,D]."""
    dt = t.dtype
    cls = _w(sd, "class_token", dt).expand(t.shape[0], -1, -1)
    return torch.cat([cls, t], dim=1) + _w(sd, "encoder.pos_embedding", dt)


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def engine_attention_form(tokens: int, head_dim: int) -> str:
    """Which attention kernel the engine launches for P.V (kernels_attn.hip: use_q32): "q32" from 289 tokens on at head dim 64, else "one-pass"
    (the attention-map inspector is always the one-pass form).  tests/test_gpu_configs.py asserts the kernel names per configuration."""
    return "q32" if head_dim == 64 and tokens > 288 else "one-pass"


def _pv_emulated(sc: torch.Tensor, v: torch.Tensor, p_dtype=None, point="p"):
    """softmax(sc) v as the engine's attention kernel evaluates it, scores sc [B,H,N,N] already scaled, v [B,H,N,dh] -> [B,H,N,dh]."""
    if engine_attention_form(sc.shape[-1], v.shape[-1]) == "q32":
        # ivit_attention_q32: numerators 2^(c s - m) with an INTEGER reference exponent m (the kernel takes ceil(c max) over the first key tile
        # a wave sees; any integer gives the same rounded numerators up to an exact power of two), rounded to 16 bits; the row sum is the sum
        # of the ROUNDED numerators
        l2 = sc * math.log2(math.e)
        pr = rnd(torch.exp2(l2 - torch.ceil(l2.amax(dim=-1, keepdim=True))), True, p_dtype, point=point)
        return (pr @ v) / pr.sum(dim=-1, keepdim=True)
    # one-pass kernel: numerators exp(s - max), rounded to 16 bits for P.V; the row sum is the sum of the UNROUNDED numerators
    e = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
    return (rnd(e, True, p_dtype, point=point) @ v) / e.sum(dim=-1, keepdim=True)


def attention_core(qkv: torch.Tensor, cfg, emulate: bool = False, p_dtype=None):
    """Scaled-dot-product attention on a stored q|k|v tensor [B,N,3D] -> (output [B,N,D] before any rounding,
    probabilities [B,H,N,N]).  emulate: the engine's evaluation - one-pass kernel: numerators e = exp(s - max) in f32/f64, row sum
    of the UNROUNDED e, P.V on the 16-bit copy of e (``p_dtype`` pins that type where the engine's policy does); 32-query tiled kernel
    (engine_attention_form): see below."""
    b, n, d3 = qkv.shape
    d = d3 // 3
    hd = cfg.head_dim
    q, k, v = qkv.split(d, dim=-1)

    def heads(t):
        return t.reshape(b, n, cfg.heads, hd).transpose(1, 2)  # [B,H,N,dh]

    q, k, v = heads(q), heads(k), heads(v)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if emulate:
        p = torch.softmax(s, dim=-1)   # (the probabilities the attention-map inspector writes: unrounded numerators over their sum)
        a = _pv_emulated(s, v, p_dtype)
    else:
        p = torch.softmax(s, dim=-1)
        a = p @ v
    return a.transpose(1, 2).reshape(b, n, d), p


def attention(h: torch.Tensor, sd, i: int, cfg, return_probs: bool = False, emulate: bool = False):
    dt = h.dtype
    pre = layer_prefix(i) + "self_attention."
    qkv = rnd(h @ _w(sd, pre + "in_proj_weight", dt, emulate, "w_qkv").t() + _w(sd, pre + "in_proj_bias", dt), emulate, point="qkv")
    a, p = attention_core(qkv, cfg, emulate)
    if emulate and _split("proj"):
        out = _split_product(a, sd[pre + "out_proj.weight"].to(dt)) + _w(sd, pre + "out_proj.bias", dt)
        return (out, p) if return_probs else out
    a = rnd(a, emulate, point="att")
    out = a @ _w(sd, pre + "out_proj.weight", dt, emulate, "w_proj").t() + _w(sd, pre + "out_proj.bias", dt)
    return (out, p) if return_probs else out


def attention_map(x: torch.Tensor, sd, i: int, cfg, emulate: bool = False) -> torch.Tensor:
    """Attention probabilities of layer i for a residual-stream input: [B,N,D] -> [B,H,N,N].  In the rounding-aware mode the q|k|v
    tensor is evaluated as the ENGINE's layer call evaluates it (LN_FOLD: LayerNorm folded into the QKV GEMM) - the map the layer itself
    attends with, which is what the `.attn` inspector and the `.with_attn` channel return since round 4."""
    dt = x.dtype
    pre = layer_prefix(i)
    if emulate and LN_FOLD:
        b, n, d = x.shape
        qkv = rnd(folded_linear(x, sd[pre + "self_attention.in_proj_weight"], sd[pre + "self_attention.in_proj_bias"],
                                sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], cfg.ln_eps, "w_qkv", _split("qkvw"), _centre(2 * i)), True, point="qkv")
        return attention_core(qkv, cfg, True)[1]
    h = rnd(layer_norm(x, _w(sd, pre + "ln_1.weight", dt), _w(sd, pre + "ln_1.bias", dt), cfg.ln_eps), emulate, point="x")
    return attention(h, sd, i, cfg, return_probs=True, emulate=emulate)[1]


# Which rounding points the rounding-aware mode (emulate=True) mirrors inside an encoder layer.  The engine's
# default on the bf16 data path folds each LayerNorm into the GEMM that consumes it (include/ivit.h:
# ivit_ln_fold): the GEMM multiplies bf16(x) by bf16(W . diag(gamma)) and applies
# rstd * (acc - mean * s) + c in its epilogue, s = row sums of the rounded W', c = W beta + b.  Same f32
# contract, different bf16 rounding points (their distance from the f32 forward is the same: 1.4e-3 per layer on
# ViT-B/16).  Tests / bench / smoke set this from Engine.ln_fold; the plain (emulate=False) forward never looks at it.
LN_FOLD = False

# The engine's centred operand copies (round 5; include/ivit.h: ivit_ln_fold_calibrate): None, or a [2 * layers, D] tensor of per-channel
# centre vectors m (row 2 i: LN1 of layer i, row 2 i + 1: LN2) - the folded GEMM then multiplies rn16(x - m) and adds W' m back:
# rstd ((rn16(x - m) + m) W'^T - mean s) + c.  Tests / bench / smoke set it from Engine.ln_centres(); only the LN_FOLD form looks at it.
LN_CENTRE = None


def _centre(site: int):
    return None if LN_CENTRE is None else LN_CENTRE[site]


# How the folded weight W' = W . diag(gamma) is prepared.  "twice" = what the ENGINE does for every non-split matrix
# (ivit_fold_ln_weights): from its 16-bit copy of W, rounded again.  "once" = from the f32 matrix, one rounding - a STUDY-ONLY
# switch of tools/f16_error_terms.py (it changes nothing for gamma = 1 and the engine does not implement it; the weight-split
# matrices of IVIT_PRECISION_F16X are folded from the f32 original, which `wsplit` below mirrors).  No test sets it.
FOLD_ROUNDING = "twice"


def folded_linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                  wpoint=None, wsplit: bool = False, centre=None) -> torch.Tensor:
    """LayerNorm(x) @ w.T + b as the engine's folded GEMM evaluates it (x: [..., D] in the compute dtype; centre: the per-channel
    vector m the engine subtracts before the 16-bit rounding of its operand copy, or None)."""
    dt = x.dtype
    if not _on(wpoint):
        wb = w.to(dt)
        wf = wb * gamma.to(dt)[None, :]
    elif wsplit:                                  # W' = hi + lo of the f32 product W . gamma (two MFMA passes over the same x)
        wb = w.to(dt)
        wf = split16(w.to(torch.float32) * gamma.to(torch.float32)[None, :]).to(dt)
    elif FOLD_ROUNDING == "once":
        wb = w.to(dt)
        wf = (w.to(torch.float32) * gamma.to(torch.float32)[None, :]).to(OPERAND_DTYPE).to(dt)   # f32 product as on the device, one rounding
    else:
        wb = w.to(torch.float32).to(OPERAND_DTYPE).to(dt)                       # the engine's 16-bit copy of W
        wf = (wb * gamma.to(dt)[None, :]).to(torch.float32).to(OPERAND_DTYPE).to(dt)   # W . diag(gamma), rounded again
    s = wf.sum(dim=1)
    c = wb @ beta.to(dt) + b.to(dt)
    mu = x.mean(dim=-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(dim=-1, keepdim=True) + eps)
    if centre is None:
        xb = rnd(x, True, point="x")
        return rstd * (xb @ wf.t() - mu * s) + c
    m = centre.to(torch.float32)
    xb = rnd((x.to(torch.float32) - m).to(dt), True, point="x")      # the engine subtracts in f32, then rounds to 16 bits
    return rstd * ((xb @ wf.t() + m.to(dt) @ wf.t()) - mu * s) + c


def _encoder_layer_fold(x: torch.Tensor, sd, i: int, cfg) -> torch.Tensor:
    dt = x.dtype
    pre = layer_prefix(i)
    b, n, d = x.shape
    hd = cfg.head_dim
    qkv = rnd(folded_linear(x, sd[pre + "self_attention.in_proj_weight"], sd[pre + "self_attention.in_proj_bias"],
                            sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], cfg.ln_eps, "w_qkv", _split("qkvw"), _centre(2 * i)), True, point="qkv")
    q, k, v = [t.reshape(b, n, cfg.heads, hd).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    a = _pv_emulated(sc, v).transpose(1, 2).reshape(b, n, d)
    if _split("proj"):
        x = x + _split_product(a, sd[pre + "self_attention.out_proj.weight"].to(dt)) + _w(sd, pre + "self_attention.out_proj.bias", dt)
    else:
        x = x + rnd(a, True, point="att") @ _w(sd, pre + "self_attention.out_proj.weight", dt, True, "w_proj").t() + _w(sd, pre + "self_attention.out_proj.bias", dt)
    u = rnd(gelu_erf(folded_linear(x, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"], sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"], cfg.ln_eps, "w_mlp1", _split("mlp1w"), _centre(2 * i + 1))), True, point="gelu")
    return x + u @ _w16(sd, pre + "mlp.3.weight", dt, True, "w_mlp2", _split("mlp2w")).t() + _w(sd, pre + "mlp.3.bias", dt)


def encoder_layer(x: torch.Tensor, sd, i: int, cfg, emulate: bool = False) -> torch.Tensor:
    """Residual-inclusive block i: [B,N,D] -> [B,N,D]."""
    if emulate and LN_FOLD:
        return _encoder_layer_fold(x, sd, i, cfg)
    dt = x.dtype
    pre = layer_prefix(i)
    h = rnd(layer_norm(x, _w(sd, pre + "ln_1.weight", dt), _w(sd, pre + "ln_1.bias", dt), cfg.ln_eps), emulate, point="x")
    x = x + attention(h, sd, i, cfg, emulate=emulate)
    h = rnd(layer_norm(x, _w(sd, pre + "ln_2.weight", dt), _w(sd, pre + "ln_2.bias", dt), cfg.ln_eps), emulate, point="x")
    m = rnd(gelu_erf(h @ _w16(sd, pre + "mlp.0.weight", dt, emulate, "w_mlp1", _split("mlp1w")).t() + _w(sd, pre + "mlp.0.bias", dt)), emulate, point="gelu")
    return x + m @ _w16(sd, pre + "mlp.3.weight", dt, emulate, "w_mlp2", _split("mlp2w")).t() + _w(sd, pre + "mlp.3.bias", dt)


# ---- fp8 data path (BASELINE config 5): the engine's policy restated (include/ivit.h, ivit_fp8_calibrate)
FP8_MAX = 448.0


def q8_act(t: torch.Tensor, scale: float) -> torch.Tensor:
    """Per-tensor static quantisation of an activation: sat_e4m3(t / scale) * scale."""
    inv = float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(scale, dtype=torch.float32))
    q = (t * inv).clamp(-FP8_MAX, FP8_MAX).to(torch.float32).to(torch.float8_e4m3fn).to(t.dtype)
    return q * scale


def q8_weight(sd, key: str, dtype) -> torch.Tensor:
    """Per-output-row quantisation of a weight matrix FROM ITS bf16 COPY (what the engine holds)."""
    w = sd[key].to(torch.float32).to(torch.bfloat16).to(torch.float32)
    amax = w.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    inv = 1.0 / scale                                   # f32, as on the device
    q = (w * inv).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).to(dtype)


def encoder_layer_fp8(x: torch.Tensor, sd, i: int, cfg, scales4, mlp_only: bool = False) -> torch.Tensor:
    """Block i on the fp8 data path: e4m3 operands for the four GEMMs (static activation scales
    s_h1, s_att, s_h2, s_u; per-row weight scales), bf16 q|k|v and P, f32/f64 everything else.
    ``mlp_only`` (IVIT_PRECISION_FP8M): the attention half on the bf16 data path with LayerNorm kernels, e4m3 for MLP up / down only."""
    s_h1, s_att, s_h2, s_u = [float(v) for v in scales4]
    dt = x.dtype
    pre = layer_prefix(i)
    b, n, d = x.shape
    hd = cfg.head_dim
    if mlp_only:
        def r16(t):
            return t.to(torch.float32).to(torch.bfloat16).to(dt)
        h = r16(layer_norm(x, _w(sd, pre + "ln_1.weight", dt), _w(sd, pre + "ln_1.bias", dt), cfg.ln_eps))
        qkv = r16(h @ r16(sd[pre + "self_attention.in_proj_weight"].to(dt)).t() + _w(sd, pre + "self_attention.in_proj_bias", dt))
        a, _ = attention_core(qkv, cfg, True, torch.bfloat16)
        x = x + r16(a) @ r16(sd[pre + "self_attention.out_proj.weight"].to(dt)).t() + _w(sd, pre + "self_attention.out_proj.bias", dt)
        h = q8_act(layer_norm(x, _w(sd, pre + "ln_2.weight", dt), _w(sd, pre + "ln_2.bias", dt), cfg.ln_eps), s_h2)
        u = q8_act(gelu_erf(h @ q8_weight(sd, pre + "mlp.0.weight", dt).t() + _w(sd, pre + "mlp.0.bias", dt)), s_u)
        return x + u @ q8_weight(sd, pre + "mlp.3.weight", dt).t() + _w(sd, pre + "mlp.3.bias", dt)
    h = q8_act(layer_norm(x, _w(sd, pre + "ln_1.weight", dt), _w(sd, pre + "ln_1.bias", dt), cfg.ln_eps), s_h1)
    qkv = rnd(h @ q8_weight(sd, pre + "self_attention.in_proj_weight", dt).t() + _w(sd, pre + "self_attention.in_proj_bias", dt), True, torch.bfloat16)
    q, k, v = [t.reshape(b, n, cfg.heads, hd).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    a = _pv_emulated(sc, v, torch.bfloat16, point=None)
    a = q8_act(a.transpose(1, 2).reshape(b, n, d), s_att)
    x = x + a @ q8_weight(sd, pre + "self_attention.out_proj.weight", dt).t() + _w(sd, pre + "self_attention.out_proj.bias", dt)
    h = q8_act(layer_norm(x, _w(sd, pre + "ln_2.weight", dt), _w(sd, pre + "ln_2.bias", dt), cfg.ln_eps), s_h2)
    u = q8_act(gelu_erf(h @ q8_weight(sd, pre + "mlp.0.weight", dt).t() + _w(sd, pre + "mlp.0.bias", dt)), s_u)
    return x + u @ q8_weight(sd, pre + "mlp.3.weight", dt).t() + _w(sd, pre + "mlp.3.bias", dt)


def fp8_calibration_scales(x: torch.Tensor, sd, cfg):
    """The calibration the engine performs, on the oracle: one bf16-emulated forward, amax / 448 of
    the four GEMM-input tensors of every layer (bf16-rounded, as the engine measures them)."""
    t = x
    for s_ in ("transform", "conv_proj", "tokens"):
        t = run_node(s_, t, sd, cfg, emulate=True) if s_ != "transform" else transform(t.to(torch.float32)).to(x.dtype)
    scales = []
    for i in range(cfg.layers):
        dt = t.dtype
        pre = layer_prefix(i)
        h1 = rnd(layer_norm(t, _w(sd, pre + "ln_1.weight", dt), _w(sd, pre + "ln_1.bias", dt), cfg.ln_eps), True)
        att_in = attention(h1, sd, i, cfg, emulate=True)   # includes out-proj; recompute the pre-projection tensor below
        # pre-projection attention output (bf16-rounded), as in attention(...)
        b, n, d = h1.shape
        qkv = rnd(h1 @ _w(sd, pre + "self_attention.in_proj_weight", dt, True).t() + _w(sd, pre + "self_attention.in_proj_bias", dt), True)
        q, k, v = [z.reshape(b, n, cfg.heads, cfg.head_dim).transpose(1, 2) for z in qkv.split(d, dim=-1)]
        sc = (q @ k.transpose(-1, -2)) / math.sqrt(cfg.head_dim)
        a = rnd(_pv_emulated(sc, v, point=None).transpose(1, 2).reshape(b, n, d), True)
        t2 = t + att_in
        h2 = rnd(layer_norm(t2, _w(sd, pre + "ln_2.weight", dt), _w(sd, pre + "ln_2.bias", dt), cfg.ln_eps), True)
        u = rnd(gelu_erf(h2 @ _w(sd, pre + "mlp.0.weight", dt, True).t() + _w(sd, pre + "mlp.0.bias", dt)), True)
        t = t2 + u @ _w(sd, pre + "mlp.3.weight", dt, True).t() + _w(sd, pre + "mlp.3.bias", dt)
        scales += [float(z.abs().max()) / FP8_MAX for z in (h1, a, h2, u)]
    return scales


def forward_fp8(x: torch.Tensor, sd, cfg, scales, keep: bool = False, mlp_only: bool = False) -> Dict[str, torch.Tensor]:
    """Whole model on the fp8 data path: patch embedding and head as in emulate=True (bf16), every
    encoder layer through encoder_layer_fp8 with the given L*4 activation scales."""
    acts: Dict[str, torch.Tensor] = {}
    t = transform(x.to(torch.float32)).to(x.dtype)
    t = tokens(conv_proj(t, sd, cfg, True), sd, cfg)
    if keep:
        acts["tokens"] = t
    for i in range(cfg.layers):
        t = encoder_layer_fp8(t, sd, i, cfg, scales[4 * i:4 * i + 4], mlp_only)
        if keep:
            acts[f"encoder.layers.{i}"] = t
    t = encoder_ln(t, sd, cfg)
    acts["cls"] = cls(t)
    acts["logits"] = acts["heads"] = heads(acts["cls"], sd, True)
    return acts


def encoder_ln(x: torch.Tensor, sd, cfg) -> torch.Tensor:
    dt = x.dtype
    return layer_norm(x, _w(sd, "encoder.ln.weight", dt), _w(sd, "encoder.ln.bias", dt), cfg.ln_eps)


def cls(x: torch.Tensor) -> torch.Tensor:
    return x[:, 0, :]


def heads(x: torch.Tensor, sd, emulate: bool = False) -> torch.Tensor:
    dt = x.dtype
    if emulate and _split("head"):
        return _split_product(x, sd["heads.head.weight"].to(dt)) + _w(sd, "heads.head.bias", dt)
    return rnd(x, emulate, point="head_in") @ _w(sd, "heads.head.weight", dt, emulate, "w_head").t() + _w(sd, "heads.head.bias", dt)


def node_suffixes(cfg) -> List[str]:
    """The chain of nodes of the plugin, in graph order (SURVEY 8(a2'))."""
    return (["transform", "conv_proj", "tokens"]
            + [f"encoder.layers.{i}" for i in range(cfg.layers)]
            + ["encoder.ln", "cls", "heads"])


def run_node(suffix: str, x: torch.Tensor, sd, cfg, emulate: bool = False) -> torch.Tensor:
    """One node on a BATCHED input (leading B axis)."""
    if suffix == "transform":
        return transform(x)
    if suffix == "preprocess":
        return preprocess(x, cfg)
    if suffix == "conv_proj":
        return conv_proj(x, sd, cfg, emulate)
    if suffix == "tokens":
        return tokens(x, sd, cfg)
    if suffix.startswith("encoder.layers.") and suffix.endswith(".attn"):
        return attention_map(x, sd, int(suffix.split(".")[-2]), cfg, emulate)
    if suffix.startswith("encoder.layers."):
        return encoder_layer(x, sd, int(suffix.rsplit(".", 1)[1]), cfg, emulate)
    if suffix == "encoder.ln":
        return encoder_ln(x, sd, cfg)
    if suffix == "cls":
        return cls(x)
    if suffix == "heads":
        return heads(x, sd, emulate)
    if suffix == "forward":
        return forward(x, sd, cfg, emulate=emulate)["logits"]
    raise KeyError(suffix)


def forward(x: torch.Tensor, sd, cfg, keep: bool = False, emulate: bool = False) -> Dict[str, torch.Tensor]:
    """Whole model, [B,3,S,S] in [0,1] -> {"logits": [B,classes], "cls": [B,D], + every node}.

    In emulate mode the transform is evaluated in float32 exactly as the engine does (its result is
    then rounded to bf16 by conv_proj), whatever the dtype of the rest."""
    acts: Dict[str, torch.Tensor] = {}
    t = x
    for s in node_suffixes(cfg):
        if emulate and s == "transform":
            t = transform(t.to(torch.float32)).to(x.dtype)
            if keep:
                acts[s] = t
            continue
        t = run_node(s, t, sd, cfg, emulate)
        if keep or s in ("cls", "heads"):
            acts[s] = t
    acts["logits"] = acts["heads"]
    return acts


# unbatched ranks of each node's input, to accept both [3,S,S] and [B,3,S,S] style tensors
INPUT_RANK = {"transform": 3, "preprocess": 3, "conv_proj": 3, "tokens": 2, "encoder.ln": 2, "cls": 2, "heads": 1,
              "forward": 3}


def input_rank(suffix: str) -> int:
    return 2 if suffix.startswith("encoder.layers.") else INPUT_RANK[suffix]   # incl. the .attn inspectors


def run_node_any(suffix: str, x: torch.Tensor, sd, cfg, emulate: bool = False) -> torch.Tensor:
    """Accepts the unbatched interactive form ([3,S,S], [N,D], [D]) or the batched one."""
    if x.dim() == input_rank(suffix):
        return run_node(suffix, x.unsqueeze(0), sd, cfg, emulate).squeeze(0)
    return run_node(suffix, x, sd, cfg, emulate)
