"""``cos`` operator: y = cos(A*x + b).  The plumbing known-answer op of the node path.

Same contract as the reference's ``main/nodes/cos.py:7-37``: params ``A`` (default 1.0) and ``b``
(default 0.0) arrive as str or number and go through ``float()``; one input "o", one output "o";
a missing input raises ``Exception("missing input: o")`` which /compute turns into HTTP 400.
"""
from typing import Dict, Mapping, Tuple

import torch

try:  # dropped into the reference tree (main/nodes/) the reference's own API is the base class
    from main.context import NodeKind
    from main.graph import Pinout
except ImportError:
    from interactive_vit_amd.context import NodeKind
    from interactive_vit_amd.graph import Pinout


class CosNode(NodeKind):
    def __init__(self):
        super().__init__("cos")

    @staticmethod
    def decode_params(params: Mapping[str, str]) -> Tuple[float, float]:
        # `in` on purpose: params=None (JSON null) must fail exactly as the reference does
        a = float(params["A"]) if "A" in params else 1.0
        b = float(params["b"]) if "b" in params else 0.0
        return a, b

    def contents(self, params: Mapping[str, str]) -> str:
        a, b = self.decode_params(params)
        return f"cos({a}x+{b})"

    def io(self, params: Mapping[str, str]) -> Dict:
        return {"ins": ["o"], "outs": ["o"]}

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        a, b = self.decode_params(params)
        x = inputs.get("o")
        if x is None:
            raise Exception("missing input: o")
        res = Pinout()
        res.set("o", torch.cos(a * x + b))
        return res


def instances():
    return [CosNode()]
