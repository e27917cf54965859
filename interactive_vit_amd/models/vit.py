"""ViT model plugin for the node-graph operator API, in the style of the reference's
``static/models/vgg16.py:10-62``: a ``Model`` subclass that overrides ``list_node_names``,
``compute``, ``contents`` and ``generate_graph_json``, plus a module-level ``instances()``.

Nodes (``<name>:`` prefix): ``transform``, ``conv_proj``, ``tokens``, ``encoder.layers.<i>``
(residual-inclusive: the server graph cannot fan out), ``encoder.ln``, ``cls``,
``heads`` - a linear chain that ends in a client-side ``category`` node - the standalone
``forward`` node (whole model in one launch sequence), the standalone ``preprocess`` node (raw image of
any size -> resize / centre crop / normalise, the classification preset of vgg16.py:40-42; use it in place
of ``transform``), and ``encoder.layers.<i>.attn`` inspectors
(``[N,D]`` residual stream in, attention probabilities ``[heads,N,N]`` out - a ``[C,H,W]`` tensor the
client's MultiView displays).  Every node takes one input "o" and gives
one output "o"; images are unbatched ``[3,S,S]`` in the interactive path, a leading batch axis is
accepted everywhere.

All arithmetic happens in ``libivit.so`` (``interactive_vit_amd.engine.Engine``, hand-written
gfx950 kernels).  There is no torch/CPU implementation behind these nodes: without the library or a
GPU, constructing the backend raises and the plugin does not register (the reference logs and skips
a plugin whose import fails, main/context.py:173-174).

The class is produced by ``make_vit_model_class(ModelBase, PinoutCls)`` so that the same code
registers against the reference's own ``main.context.Model`` (drop-in) or
against this package's restatement of it.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from ..graphjson import chain_graph
from ..vit_config import VARIANTS, VitConfig
from ..weights import init_weights


class VitParameters(torch.nn.Module):
    """Parameter container with torchvision ``VisionTransformer`` state-dict names.

    It has no ``forward``: the ``Model`` base class only needs an ``nn.Module`` to own the weights
    (reference main/context.py:39-41); the forward lives on the GPU.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor]):
        super().__init__()
        for key, value in state_dict.items():
            *path, leaf = key.split(".")
            mod = self
            for part in path:
                if not hasattr(mod, part):
                    mod.add_module(part, torch.nn.Module())
                mod = getattr(mod, part)
            mod.register_parameter(leaf, torch.nn.Parameter(value.detach().clone(), requires_grad=False))

    def forward(self, *args, **kwargs):
        raise RuntimeError("VitParameters holds weights only; the forward runs in libivit.so on the GPU")


def node_suffixes(cfg: VitConfig) -> List[str]:
    return (["transform", "conv_proj", "tokens"] + [f"encoder.layers.{i}" for i in range(cfg.layers)]
            + ["encoder.ln", "cls", "heads"])


def default_categories(classes: int, path: Optional[str] = None) -> List[str]:
    """Class labels for the ``category`` node: one per line from a local file, else placeholders."""
    path = path or os.environ.get("IVIT_CATEGORIES")
    if path and os.path.exists(path):
        with open(path) as f:
            cats = [ln.strip() for ln in f if ln.strip()]
        if len(cats) == classes:
            return cats
    return [f"class {i}" for i in range(classes)]


def make_vit_model_class(ModelBase, PinoutCls):
    """Binds the plugin to a concrete operator API (the reference's or this package's)."""

    class VitModel(ModelBase):
        def __init__(self, cfg: VitConfig, backend, name: Optional[str] = None,
                     categories: Optional[Sequence[str]] = None):
            self.cfg = cfg
            self.backend = backend          # object with run_node(suffix, tensor) -> tensor
            self.categories = list(categories) if categories is not None else default_categories(cfg.classes)
            self._suffixes = node_suffixes(cfg)
            super().__init__(backend.module(), name or cfg.name)

        # -- node set ----------------------------------------------------------------------
        def chain_node_names(self) -> List[str]:
            return [self.prefix() + s for s in self._suffixes]

        def attn_node_names(self) -> List[str]:
            return [f"{self.prefix()}encoder.layers.{i}.attn" for i in range(self.cfg.layers)]

        def with_attn_node_names(self) -> List[str]:
            # the layer node with its attention map as a SECOND output channel ("o" continues the chain, "attn" is shipped by Response
            # like every channel of every node, main/message.py:80-83) - SURVEY 8(f) row 4 as written.  The reference's ModelNode drops
            # `params` (main/context.py:119-129), so the two-channel form is a node of its own rather than a parameter of the layer node.
            return [f"{self.prefix()}encoder.layers.{i}.with_attn" for i in range(self.cfg.layers)]

        def list_node_names(self) -> List[str]:
            # the chain, the fused whole model, and one attention-map inspector per encoder layer
            # ([N,D] -> [heads,N,N]; put it in place of layer i at the end of a shorter chain: the
            # server graph cannot fan out)
            # ... and `preprocess`, the raw-image front end (any [3,H,W] -> what `transform` gives for [3,S,S]):
            # put it in place of `transform` when the image does not come from a client-side Resize node
            return (self.chain_node_names() + [self.prefix() + "forward", self.prefix() + "preprocess"] + self.attn_node_names()
                    + self.with_attn_node_names())

        def generate_graph_json(self) -> Dict:
            """Chain graph in the client's schema (graph.js:700-758), laid out exactly like
            ``Model.generate_graph_json`` (context.py:55-73) and closed by a ``category`` node the
            way vgg16.py:16-29 does."""
            return chain_graph(self.chain_node_names(), tail={"kind": "category", "cats": self.categories})

        # -- operator interface --------------------------------------------------------------
        def compute(self, node_name: str, pinin):
            x = pinin.get("o")
            assert x is not None
            suffix = node_name.removeprefix(self.prefix())
            if node_name in self.with_attn_node_names():   # two output channels: "o" (the layer), "attn" ([heads,N,N])
                with torch.no_grad():
                    ys = self.backend.run_node_multi(suffix, x)
                out = PinoutCls()
                for channel in ("o", "attn"):
                    assert isinstance(ys[channel], torch.Tensor)
                    out.set(channel, ys[channel])
                return out
            if suffix not in ("forward", "preprocess") and suffix not in self._suffixes and node_name not in self.attn_node_names():
                raise KeyError(node_name)
            with torch.no_grad():
                y = self.backend.run_node(suffix, x)
            assert isinstance(y, torch.Tensor)
            out = PinoutCls()
            out.set("o", y)
            return out

        def contents(self, node_name: str) -> str:
            suffix = node_name.removeprefix(self.prefix())
            c = self.cfg
            detail = {
                "transform": "normalise (ImageNet mean/std)",
                "preprocess": f"resize {(c.image * 256 + 112) // 224} (antialiased) &rarr; centre crop {c.image} &rarr; normalise",
                "conv_proj": f"patch embed {c.patch}x{c.patch} &rarr; [{c.patches},{c.dim}]",
                "tokens": f"[CLS] + position &rarr; [{c.tokens},{c.dim}]",
                "encoder.ln": "LayerNorm",
                "cls": f"token 0 &rarr; [{c.dim}]",
                "heads": f"Linear &rarr; [{c.classes}]",
                "forward": f"whole model &rarr; [{c.classes}]",
            }.get(suffix, f"attention map &rarr; [{c.heads},{c.tokens},{c.tokens}]" if suffix.endswith(".attn")
                  else (f"MHSA({c.heads} heads) + MLP({c.mlp}); second channel attn &rarr; [{c.heads},{c.tokens},{c.tokens}]" if suffix.endswith(".with_attn")
                        else f"MHSA({c.heads} heads) + MLP({c.mlp})"))
            return f"<p>{node_name}</p> <p>{detail}</p>"

        def io(self, node_name: str) -> Dict:
            if node_name in self.with_attn_node_names():
                return {"ins": ["o"], "outs": ["o", "attn"]}
            return {"ins": ["o"], "outs": ["o"]}

    return VitModel


class HipBackend:
    """The product backend: owns an ``Engine`` on one GPU.  Raises if the GPU path is unavailable."""

    def __init__(self, cfg: VitConfig, state_dict: Dict[str, torch.Tensor], device: int = 0, max_batch: int = 1,
                 precision: str = "bf16", check_ln_fold: bool = True, calibration_images: Optional[torch.Tensor] = None,
                 ln_fold_threshold: float = 0.5):
        from ..engine import Engine  # raises when libivit.so is missing
        from ..weights import synthetic_images
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible (torch.cuda.is_available() is False); the ViT nodes have no CPU path")
        self.cfg = cfg
        self._module = VitParameters(state_dict)
        self.engine = Engine(cfg, state_dict, device=device, max_batch=max_batch, precision=precision)
        # A plugin takes whatever checkpoint it is given (reference static/models/vgg16.py:12-14).  The LayerNorm fold
        # of the engine is only as accurate as the unfolded form while the 16-bit copies of the residual-stream rows it multiplies
        # are small against the rows' spread, a property of the weights: one calibration forward on sample images takes the
        # per-channel means of every LayerNorm input as the centre vectors of those copies (real checkpoints' offsets and outlier
        # channels are constant across rows), measures what is left, and lets the engine keep or drop the fold.
        # `calibration_images` ([B,3,S,S] in [0,1], B <= max_batch): real sample pictures of the deployment when the operator has some
        # (a real checkpoint's residual stream depends on its inputs); else two seeded synthetic images.  `ln_fold_threshold`: the
        # largest |mean| / std the fold is kept for (engine default 0.5); the measured ratio is kept in `ln_fold_ratio`.
        self.ln_fold_ratio = None
        if check_ln_fold and precision not in ("fp8", "fp8m") and self.engine.ln_fold:
            if calibration_images is None:
                # say so - uniform-noise pictures exercise the weights' own statistics, not a deployment's inputs
                import logging
                logging.getLogger(__name__).warning(
                    "%s: LayerNorm-fold guard calibrated on %d seeded SYNTHETIC images (no calibration_images given); pass real sample "
                    "pictures to HipBackend / build_plugins for a checkpoint whose residual stream depends on its inputs", cfg.name, min(2, max_batch))
            images = calibration_images if calibration_images is not None else synthetic_images(min(2, max_batch), cfg, seed=7)
            if images.dim() == 3:
                images = images.unsqueeze(0)
            self.ln_fold_ratio = self.engine.calibrate_ln_fold(images[:max_batch], threshold=ln_fold_threshold)
        # The e4m3 data paths need their static activation scales before any layer node can run (engine: "fp8 engine is not calibrated"): the plugin
        # calibrates them here, on the operator's sample pictures when there are some (nothing in the plugin path ever did, so every layer node
        # of an IVIT_PRECISION=fp8 / fp8m deployment returned that message).
        self.fp8_scales = None
        if precision in ("fp8", "fp8m"):
            import logging
            if calibration_images is None:
                logging.getLogger(__name__).warning(
                    "%s: e4m3 activation scales calibrated on %d seeded SYNTHETIC images (no calibration_images given); pass real sample pictures to "
                    "HipBackend / build_plugins", cfg.name, min(2, max_batch))
            images = calibration_images if calibration_images is not None else synthetic_images(min(2, max_batch), cfg, seed=7)
            if images.dim() == 3:
                images = images.unsqueeze(0)
            self.fp8_scales = self.engine.calibrate_fp8(images[:max_batch])

    def module(self) -> torch.nn.Module:
        return self._module

    def run_node(self, suffix: str, x: torch.Tensor) -> torch.Tensor:
        return self.engine.run_node(suffix, x)

    def run_node_multi(self, suffix: str, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        return self.engine.run_node_multi(suffix, x)


def build_plugins(ModelBase, PinoutCls, variants: Sequence[str] = ("vit_b_16",), device: int = 0,
                  max_batch: int = 1, seed: int = 0, state_dicts: Optional[Dict[str, Dict[str, torch.Tensor]]] = None,
                  precision: str = "bf16", calibration_images: Optional[torch.Tensor] = None, ln_fold_threshold: float = 0.5):
    """What a plugin file's ``instances()`` returns: one registered model per variant.

    Weights: ``state_dicts[name]`` when given (e.g. loaded from a local safetensors file with
    torchvision key names), else the seeded synthetic initialisation (no network).
    """
    cls = make_vit_model_class(ModelBase, PinoutCls)
    models = []
    for v in variants:
        cfg = VARIANTS[v]
        sd = (state_dicts or {}).get(v) or init_weights(cfg, seed=seed, mode="spec")
        models.append(cls(cfg, HipBackend(cfg, sd, device=device, max_batch=max_batch, precision=precision,
                                          calibration_images=calibration_images, ln_fold_threshold=ln_fold_threshold)))
    return models


def instances():
    """Entry point when this file itself is scanned as a plugin (``scan_nodes``)."""
    try:  # inside the reference tree its own operator API is the base class
        from main.context import Model as ModelBase
        from main.graph import Pinout as PinoutCls
    except ImportError:
        from ..context import Model as ModelBase
        from ..graph import Pinout as PinoutCls
    variants = tuple(v for v in os.environ.get("IVIT_VARIANTS", "vit_b_16").split(",") if v)
    # IVIT_WEIGHTS = "<path>" (one variant) or "vit_b_16=<path>,vit_l_16_384=<path>": local checkpoints in torchvision key names
    # (weights.load_state_dict_file); variants without one get the seeded synthetic initialisation.  IVIT_CATEGORIES = label file.
    state_dicts = {}
    spec = os.environ.get("IVIT_WEIGHTS", "")
    if spec:
        from ..weights import load_state_dict_file
        for item in spec.split(","):
            # "<variant>=<path>": only a known variant name in front of the FIRST '=' is a name (a path may contain '=')
            head, sep, tail = item.partition("=")
            name, path = (head, tail) if (sep and head in VARIANTS) else (variants[0], item)
            if name not in variants:
                raise ValueError(f"IVIT_WEIGHTS names variant '{name}', which IVIT_VARIANTS ({','.join(variants)}) does not serve")
            state_dicts[name] = load_state_dict_file(path, VARIANTS[name])
    return build_plugins(ModelBase, PinoutCls, variants, state_dicts=state_dicts,
                         device=int(os.environ.get("IVIT_DEVICE", "0")),
                         max_batch=int(os.environ.get("IVIT_MAX_BATCH", "1")),
                         precision=os.environ.get("IVIT_PRECISION", "bf16"))
