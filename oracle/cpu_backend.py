"""ORACLE - test infrastructure, not product code.

CPU backend for the ViT plugin class: runs each node with the pure-torch restatement in
``vit_oracle.py``.  Plugged into ``make_vit_model_class`` it gives the "CPU node-graph forward" -
the same structure the reference would execute (main/context.py:143-147 -> :79-88, one torch CPU
call per node) - which is (a) the parity oracle of the GPU nodes and (b) bench.py's cpu_baseline.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import vit_oracle


class _Params(torch.nn.Module):
    def __init__(self, sd: Dict[str, torch.Tensor]):
        super().__init__()
        self.sd = sd

    def forward(self, x):  # never used: the plugin overrides compute for every node
        raise RuntimeError("node-by-node only")


class OracleBackend:
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], dtype=torch.float32):
        self.cfg = cfg
        self.sd = {k: v.to(dtype) for k, v in state_dict.items()}
        self.dtype = dtype
        self._module = _Params(self.sd)

    def module(self) -> torch.nn.Module:
        return self._module

    def run_node(self, suffix: str, x: torch.Tensor) -> torch.Tensor:
        y = vit_oracle.run_node_any(suffix, x.to(self.dtype), self.sd, self.cfg)
        return y.to(torch.float32)

    def run_node_multi(self, suffix: str, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """`encoder.layers.<i>.with_attn`: the layer node plus its attention map as a second channel."""
        if suffix.endswith(".with_attn"):
            layer = suffix[:-len(".with_attn")]
            return {"o": self.run_node(layer, x), "attn": self.run_node(layer + ".attn", x)}
        return {"o": self.run_node(suffix, x)}
