"""Operator/plugin API and registry: the drop-in boundary of the hot path.

Mirrors the reference's ``main/context.py``: ``NodeKind`` (:16-36), ``Model`` (:38-112),
``ModelNode`` (:114-129), ``Context`` (:132-147), the process-wide ``context()`` singleton
(:149-152) and ``scan_nodes`` plugin discovery (:154-176).  Same names, argument meaning and error
behaviour, so an operator written for the reference registers here unchanged.

Deliberate, documented differences (none observable through ``/compute``):

* the reference reads ``django.conf.settings.BASE_DIR``; here the base directory is explicit
  (``set_base_dir`` / ``IVIT_BASE_DIR``), Django is not a dependency of the hot path;
* ``scan_nodes`` is called by the embedding application, not as an import side effect (ref :176).
"""
from __future__ import annotations

import importlib.util
import json
import logging
import math
import os
import sys
import threading
from typing import Dict, Iterable, List, Mapping, Optional
from urllib.parse import urlencode

import torch

from .graph import Graph, Pinout

logger = logging.getLogger(__name__)

_base_dir: Optional[str] = os.environ.get("IVIT_BASE_DIR")


def set_base_dir(path: str) -> None:
    """Directory that contains ``main/nodes``, ``static/models`` and ``static/graphs``."""
    global _base_dir
    _base_dir = os.fspath(path)


def base_dir() -> str:
    if _base_dir is None:
        raise Exception("base directory not configured: call set_base_dir() or set IVIT_BASE_DIR")
    return _base_dir


class NodeKind:
    """One server-side operator.  Subclasses implement ``io`` and ``compute`` (ref :16-36)."""

    def __init__(self, name: str):
        self.name = name

    def get_name(self) -> str:
        return self.name

    def contents(self, params: Mapping[str, str]) -> str:
        return self.name + "?" + urlencode(params)

    def io(self, params: Mapping[str, str]) -> Dict:
        raise Exception(f"TODO: implement Node.io() for {self.name}")

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        raise Exception(f"TODO: implement Node.compute() for {self.name}")

    def register(self, ctx: "Context") -> None:
        ctx.register(self)


def _is_leaf(sub: torch.nn.Module) -> bool:
    it = sub.named_modules()
    next(it)  # the module itself
    return next(it, None) is None


class Model:
    """Exposes an ``nn.Module`` as a family of nodes named ``<model>:<dotted path>`` (ref :38-112).

    Every LEAF sub-module becomes a node (ref :44-47).  Plugins override ``list_node_names``,
    ``compute``, ``contents``, ``io`` and ``generate_graph_json`` the way the reference's
    ``static/models/vgg16.py`` does.
    """

    def __init__(self, model: torch.nn.Module, name: str):
        self.model = model
        self.model.eval()
        self.name = name
        self.node_names: List[str] = [
            self.prefix() + path for path, sub in self.model.named_modules() if _is_leaf(sub)
        ]

    def get_name(self) -> str:
        return self.name

    def prefix(self) -> str:
        return self.name + ":"

    def list_node_names(self) -> List[str]:
        return self.node_names

    def generate_graph_json(self) -> Dict:
        """Client-format graph: a linear chain on channel "o", floor(sqrt(n))-wide grid, 200 px pitch."""
        names = self.list_node_names()
        width = int(math.sqrt(len(names)))
        nodes, edges = [], []
        for i, name in enumerate(names):
            nodes.append({
                "instance": {"kind": "net_node", "endpoint": f"{name}", "params": {}},
                "pos": {"x": (i % width) * 200, "y": int(i / width) * 200},
            })
            if i:
                edges.append({"in_port": {"node": i - 1, "channel": "o"},
                              "out_port": {"node": i, "channel": "o"}})
        return {"nodes": nodes, "edges": edges}

    def compute(self, node_name: str, pinin: Pinout) -> Pinout:
        with torch.no_grad():
            sub = self.model.get_submodule(node_name.removeprefix(self.prefix()))
            x = pinin.get("o")
            assert x is not None
            res = sub(x)
            assert isinstance(res, torch.Tensor)
            return Pinout({"o": res})

    def contents(self, node_name: str) -> str:
        sub = self.model.get_submodule(node_name.removeprefix(self.prefix()))
        return f"<p>{node_name}</p> <p>{sub._get_name()}</p>"

    def io(self, node_name: str) -> Dict:
        return {"ins": ["o"], "outs": ["o"]}

    def register(self, ctx: "Context") -> None:
        # ref :98-108: first registration writes static/graphs/<name>.json if it is absent
        graph_path = os.path.join(base_dir(), "static/graphs/" + self.name + ".json")
        if not os.path.exists(graph_path):
            try:
                with open(graph_path, "w") as f:
                    f.write(json.dumps(self.generate_graph_json()))
                logger.info("generated graph %s", graph_path)
            except Exception as e:
                logger.error("could not generate graph %s: %s", graph_path, str(e))
        for node_name in self.list_node_names():
            ModelNode(self, node_name).register(ctx)


class ModelNode(NodeKind):
    """Adapter: one node of a ``Model`` seen through the ``NodeKind`` interface (ref :114-129)."""

    def __init__(self, parent: Model, name: str):
        super().__init__(name)
        self.parent = parent

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        return self.parent.compute(self.get_name(), inputs)

    def contents(self, params: Mapping[str, str]) -> str:
        return self.parent.contents(self.get_name())

    def io(self, params: Mapping[str, str]) -> Dict:
        return self.parent.io(self.get_name())


class Context:
    """Registry of operators + the per-request executor (ref :132-147)."""

    def __init__(self) -> None:
        self.nodes: Dict[str, NodeKind] = {}

    def register(self, node: NodeKind) -> None:
        logger.info("Registered node: '%s'", node.get_name())
        self.nodes[node.get_name()] = node

    def get_node(self, name: str) -> NodeKind:
        return self.nodes[name]  # unknown endpoint -> KeyError -> HTTP 400, as in the reference

    def compute(self, graph: Graph) -> None:
        for n in graph.order():
            n.set_pinout(self.get_node(n.name).compute(n.params, n.get_pinin()))


_instance = Context()
_instance_lock = threading.Lock()


def context() -> Context:
    return _instance


def scan_nodes(dirs: Iterable[str]) -> None:
    """Import every ``*.py`` under ``<base>/<dir>`` and register what its ``instances()`` returns.

    As in the reference (:154-176) a module is entered into ``sys.modules`` under its bare file
    stem, and a plugin that fails to import or register is logged and skipped.
    """
    with _instance_lock:
        for subdir in dirs:
            full_dir = os.path.join(base_dir(), subdir)
            for file in os.listdir(full_dir):
                path = os.path.join(full_dir, file)
                if not os.path.isfile(path) or not path.endswith(".py"):
                    continue
                stem = os.path.splitext(os.path.basename(path))[0]
                try:
                    spec = importlib.util.spec_from_file_location(stem, path)
                    module = importlib.util.module_from_spec(spec)
                    sys.modules[stem] = module
                    spec.loader.exec_module(module)
                    for inst in module.instances():
                        inst.register(context())
                except Exception as err:
                    logger.info("Could not register '%s': %s", path, str(err))
