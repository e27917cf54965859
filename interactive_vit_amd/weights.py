"""Seeded synthetic ViT weights (there is no network for checkpoints).

Key names follow torchvision's ``VisionTransformer`` state-dict so that a local checkpoint in
that layout can be loaded later without renaming.

``mode="spec"``  - the benchmark initialisation of SURVEY 8(d): matrices and position embedding
                   ~ N(0, 0.02^2) truncated at 2 sigma, biases and class token 0, LN gains 1.
``mode="rich"``  - same matrices, but non-trivial biases, LN gains/offsets and class token, so
                   that parity tests exercise every term of every epilogue.
"""
from __future__ import annotations

import hashlib
from typing import Dict

import torch

from .vit_config import VitConfig


def layer_prefix(i: int) -> str:
    return f"encoder.layers.encoder_layer_{i}."


def weight_shapes(cfg: VitConfig) -> Dict[str, tuple]:
    d, m, p = cfg.dim, cfg.mlp, cfg.patch
    shapes = {
        "conv_proj.weight": (d, 3, p, p),
        "conv_proj.bias": (d,),
        "class_token": (1, 1, d),
        "encoder.pos_embedding": (1, cfg.tokens, d),
    }
    for i in range(cfg.layers):
        pre = layer_prefix(i)
        shapes.update({
            pre + "ln_1.weight": (d,), pre + "ln_1.bias": (d,),
            pre + "self_attention.in_proj_weight": (3 * d, d),
            pre + "self_attention.in_proj_bias": (3 * d,),
            pre + "self_attention.out_proj.weight": (d, d),
            pre + "self_attention.out_proj.bias": (d,),
            pre + "ln_2.weight": (d,), pre + "ln_2.bias": (d,),
            pre + "mlp.0.weight": (m, d), pre + "mlp.0.bias": (m,),
            pre + "mlp.3.weight": (d, m), pre + "mlp.3.bias": (d,),
        })
    shapes.update({
        "encoder.ln.weight": (d,), "encoder.ln.bias": (d,),
        "heads.head.weight": (cfg.classes, d), "heads.head.bias": (cfg.classes,),
    })
    return shapes


def init_weights(cfg: VitConfig, seed: int = 0, mode: str = "spec") -> Dict[str, torch.Tensor]:
    assert mode in ("spec", "rich")
    gen = torch.Generator(device="cpu").manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        is_gain = name.endswith(("ln_1.weight", "ln_2.weight", "ln.weight"))
        is_matrix = len(shape) >= 2 and name != "class_token"
        if is_matrix:
            t = torch.empty(shape, dtype=torch.float32)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=0.02, a=-0.04, b=0.04, generator=gen)
        elif mode == "spec":
            t = torch.ones(shape) if is_gain else torch.zeros(shape)
        else:
            r = torch.randn(shape, generator=gen, dtype=torch.float32)
            t = 1.0 + 0.1 * r if is_gain else 0.05 * r
        sd[name] = t.contiguous()
    return sd


def realistic_statistics_weights(cfg: VitConfig, seed: int = 21, hot=(2.0, -2.5, 3.0, 4.0)) -> Dict[str, torch.Tensor]:
    """A random weight set with the STATISTICS of a trained checkpoint rather than N(0, 0.02^2) (no checkpoint can be fetched here): every channel of
    the position embedding offset by 0.6 (row mean of the residual stream ~ 3 sigma), a few 50 ... 100 sigma outlier channels, LayerNorm gains
    0.3 ... 3 and offsets ~ +-0.5.  What the LayerNorm-fold calibration (include/ivit.h: ivit_ln_fold_calibrate) is tested and benchmarked on
    (tests/test_gpu_configs.py, bench.py --weights realistic)."""
    sd = init_weights(cfg, seed=seed, mode="rich")
    g = torch.Generator().manual_seed(seed + 100)
    pos = sd["encoder.pos_embedding"]
    pos += 0.6
    idx = torch.randperm(cfg.dim, generator=g)[:len(hot)]
    pos[..., idx] += torch.tensor(hot)
    for i in range(cfg.layers):
        for ln in ("ln_1", "ln_2"):
            sd[f"{layer_prefix(i)}{ln}.weight"] = torch.exp(torch.randn(cfg.dim, generator=g) * 0.6).clamp(0.3, 3.0)
            sd[f"{layer_prefix(i)}{ln}.bias"] = torch.randn(cfg.dim, generator=g) * 0.25
    return sd


def load_state_dict_file(path: str, cfg: VitConfig) -> Dict[str, torch.Tensor]:
    """A LOCAL checkpoint in torchvision ``VisionTransformer`` key names -> the f32 state dict ``ivit_set_weight`` takes.

    The reference's model plugin loads real weights (static/models/vgg16.py:12-14, a network download); there is no network
    here, so the file is whatever the operator provides: ``.safetensors`` (read with the safetensors package, no pickle) or a
    ``torch.save`` state dict (``weights_only=True``).  Tensors may be f32 / bf16 / f16 and non-contiguous; every one is
    checked against ``weight_shapes(cfg)`` (a conv_proj weight may come flattened [D, 3 p p]; a timm-style ``[1, N, D]`` /
    ``[N, D]`` position embedding is accepted) and converted to contiguous f32.  Unknown extra keys are ignored (e.g. the
    ``heads.pre_logits`` of some checkpoints would change the model: those raise)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        raw = load_file(path, device="cpu")
    else:
        raw = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(raw, dict) and "state_dict" in raw and isinstance(raw["state_dict"], dict):
            raw = raw["state_dict"]
    if not isinstance(raw, dict):
        raise ValueError(f"{path}: not a state dict")
    if any(k.startswith("heads.pre_logits") for k in raw):
        raise ValueError(f"{path}: checkpoint has a pre_logits head, which this ViT definition (SURVEY App. B) does not have")
    shapes = weight_shapes(cfg)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        if name not in raw:
            raise KeyError(f"{path}: missing tensor '{name}' for {cfg.name}")
        t = raw[name]
        if not torch.is_floating_point(t):
            raise TypeError(f"{path}: '{name}' has dtype {t.dtype}")
        t = t.detach().to(torch.float32)
        if tuple(t.shape) != tuple(shape):
            # Only the two documented layouts are re-viewed; anything else with the right element count (a transposed
            # [D, Mlp] mlp.0.weight, a [D, 3D] in_proj) would run and give finite, wrong logits - refuse it by name.
            got = tuple(t.shape)
            d, p = cfg.dim, cfg.patch
            allowed = ((name == "conv_proj.weight" and got == (d, 3 * p * p))
                       or (name == "encoder.pos_embedding" and got == (cfg.tokens, d))
                       or (name == "class_token" and got in ((d,), (1, d))))
            if not allowed:
                raise ValueError(f"{path}: '{name}' has shape {got}, expected {tuple(shape)}")
            t = t.reshape(shape)
        sd[name] = t.contiguous()
    return sd


def state_digest(sd: Dict[str, torch.Tensor]) -> str:
    """sha256 over names and raw float32 bytes, in key order - pins a seeded state dict."""
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().contiguous().numpy().tobytes())
    return h.hexdigest()


def synthetic_images(batch: int, cfg: VitConfig, seed: int = 1234, device: str = "cpu") -> torch.Tensor:
    """x ~ U[0,1) float32 [B,3,S,S] - the browser's image range (img_source_node.js:18-20)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    return torch.rand((batch, 3, cfg.image, cfg.image), generator=gen, dtype=torch.float32, device=device)
