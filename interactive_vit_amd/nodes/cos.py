"""``cos`` operator: y = cos(A*x + b).  The plumbing known-answer op of the node path.

Same contract as the reference's ``main/nodes/cos.py:7-37``: params ``A`` (default 1.0) and ``b``
(default 0.0) arrive as str or number and go through ``float()``; one input "o", one output "o";
a missing input raises ``Exception("missing input: o")`` which /compute turns into HTTP 400.

Written as one instance of a small family - an elementwise function applied to an affine map of the
input - so that further known-answer operators are one line each.
"""
from typing import Callable, Dict, Mapping, Tuple

import torch

try:  # dropped into the reference tree (main/nodes/) the reference's own API is the base class
    from main.context import NodeKind
    from main.graph import Pinout
except ImportError:
    from interactive_vit_amd.context import NodeKind
    from interactive_vit_amd.graph import Pinout

CHANNEL = "o"
AFFINE_DEFAULTS = (("A", 1.0), ("b", 0.0))   # parameter name -> value when the request leaves it out


def affine_params(params: Mapping[str, str]) -> Tuple[float, ...]:
    """``float()`` of every affine parameter the request carries, defaults for the rest.  The membership
    test is on ``params`` itself on purpose: ``params=None`` (JSON null) fails here exactly as in the
    reference (TypeError -> HTTP 400)."""
    return tuple(float(params[key]) if key in params else default for key, default in AFFINE_DEFAULTS)


class AffineElementwise(NodeKind):
    """y = fn(A * x + b) on channel "o"."""

    def __init__(self, name: str, fn: Callable[[torch.Tensor], torch.Tensor]):
        super().__init__(name)
        self.fn = fn

    def decode_params(self, params: Mapping[str, str]) -> Tuple[float, float]:
        return affine_params(params)

    def contents(self, params: Mapping[str, str]) -> str:
        scale, shift = affine_params(params)
        return f"{self.name}({scale}x+{shift})"

    def io(self, params: Mapping[str, str]) -> Dict:
        return {"ins": [CHANNEL], "outs": [CHANNEL]}

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        scale, shift = affine_params(params)
        operand = inputs.get(CHANNEL)
        if operand is None:
            raise Exception(f"missing input: {CHANNEL}")
        result = Pinout()
        result.set(CHANNEL, self.fn(scale * operand + shift))
        return result


class CosNode(AffineElementwise):
    def __init__(self):
        super().__init__("cos", torch.cos)


def instances():
    return [CosNode()]
