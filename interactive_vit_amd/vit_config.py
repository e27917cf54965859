"""ViT variants served by the engine (BASELINE.json configs)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict


@dataclass(frozen=True)
class VitConfig:
    name: str
    image: int        # S: input is [3,S,S]
    patch: int        # p
    dim: int          # D
    heads: int        # H
    layers: int       # L
    mlp: int          # M
    classes: int = 1000
    ln_eps: float = 1e-6

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def patches(self) -> int:          # Np
        return self.grid * self.grid

    @property
    def tokens(self) -> int:           # N = Np + 1 (class token first)
        return self.patches + 1

    @property
    def patch_k(self) -> int:          # K = 3 p^2, unfold order k = c*p^2 + ky*p + kx
        return 3 * self.patch * self.patch

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    def macs_per_image(self) -> int:
        """Algorithmic multiply-accumulates of one forward; FLOPs = 2x this."""
        n, d, m = self.tokens, self.dim, self.mlp
        per_layer = 3 * n * d * d + 2 * n * n * d + n * d * d + 2 * n * d * m
        return self.patches * self.patch_k * d + self.layers * per_layer + d * self.classes

    def flops_per_image(self) -> int:
        return 2 * self.macs_per_image()

    def param_count(self) -> int:
        d, m = self.dim, self.mlp
        per_layer = 2 * 2 * d + (3 * d * d + 3 * d) + (d * d + d) + (d * m + m) + (m * d + d)
        return (self.patch_k * d + d) + d + self.tokens * d + self.layers * per_layer \
            + 2 * d + (d * self.classes + self.classes)


VARIANTS: Dict[str, VitConfig] = {
    "vit_ti_16": VitConfig("vit_ti_16", 224, 16, 192, 3, 12, 768),
    "vit_b_16": VitConfig("vit_b_16", 224, 16, 768, 12, 12, 3072),
    "vit_l_16_384": VitConfig("vit_l_16_384", 384, 16, 1024, 16, 24, 4096),
    "vit_h_14": VitConfig("vit_h_14", 224, 14, 1280, 16, 32, 5120),
}

# ImageNet statistics of the `transform` node (what torchvision's ViT presets normalise with)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def test_config(name: str = "vit_test", image: int = 64, patch: int = 16, dim: int = 128,
                heads: int = 2, layers: int = 2, mlp: int = 256, classes: int = 40) -> VitConfig:
    """Small shapes for parity tests (N = 17 tokens: exercises every ragged-tile path)."""
    return VitConfig(name, image, patch, dim, heads, layers, mlp, classes)
