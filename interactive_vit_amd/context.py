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
import os
import sys
import threading
from typing import Dict, Iterable, List, Mapping, Optional
from urllib.parse import urlencode

import torch

from .graph import Graph, Pinout
from .graphjson import chain_graph

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
    """One server-side operator (reference :16-36).  Subclasses provide ``io`` and ``compute``; the
    defaults raise, ``contents`` describes the node by its name and url-encoded parameters."""

    def __init__(self, name: str):
        self.name = name

    def get_name(self) -> str:
        return self.name

    def register(self, ctx: "Context") -> None:
        ctx.register(self)

    def contents(self, params: Mapping[str, str]) -> str:
        return f"{self.name}?{urlencode(params)}"

    def io(self, params: Mapping[str, str]) -> Dict:
        raise Exception(f"TODO: implement Node.io() for {self.name}")

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        raise Exception(f"TODO: implement Node.compute() for {self.name}")


SINGLE_IO = {"ins": ["o"], "outs": ["o"]}      # every model node: one tensor in on "o", one out on "o" (:94-96)


def leaf_paths(module: torch.nn.Module) -> List[str]:
    """Dotted paths of the sub-modules that have no children, in registration order (reference :44-47)."""
    parents = {path.rpartition(".")[0] for path, _ in module.named_modules() if path}
    return [path for path, _ in module.named_modules() if path not in parents and (path or not parents)]


class Model:
    """An ``nn.Module`` exposed as a family of nodes ``<model>:<dotted path>``, one per leaf sub-module
    (reference :38-112).  Plugins override ``list_node_names``, ``compute``, ``contents``, ``io`` and
    ``generate_graph_json`` the way the reference's ``static/models/vgg16.py`` does."""

    def __init__(self, model: torch.nn.Module, name: str):
        self.name = name
        self.model = model.eval()
        self.node_names: List[str] = [self.prefix() + path for path in leaf_paths(self.model)]

    # -- identity ---------------------------------------------------------------------------------
    def get_name(self) -> str:
        return self.name

    def prefix(self) -> str:
        return self.name + ":"

    def list_node_names(self) -> List[str]:
        return self.node_names

    def _submodule(self, node_name: str) -> torch.nn.Module:
        return self.model.get_submodule(node_name.removeprefix(self.prefix()))

    # -- operator interface, per node -----------------------------------------------------------------
    def compute(self, node_name: str, pinin: Pinout) -> Pinout:
        operand = pinin.get("o")
        assert operand is not None
        with torch.no_grad():
            result = self._submodule(node_name)(operand)
        assert isinstance(result, torch.Tensor)      # tuple-returning modules are not nodes (:85)
        return Pinout({"o": result})

    def contents(self, node_name: str) -> str:
        return f"<p>{node_name}</p> <p>{self._submodule(node_name)._get_name()}</p>"

    def io(self, node_name: str) -> Dict:
        return dict(SINGLE_IO)

    # -- registration ---------------------------------------------------------------------------------
    def generate_graph_json(self) -> Dict:
        return chain_graph(self.list_node_names())

    def _write_graph_file_once(self) -> None:
        """``static/graphs/<name>.json`` is written on first registration only (reference :98-108); a
        failure is logged, never fatal."""
        target = os.path.join(base_dir(), "static/graphs/" + self.name + ".json")
        if os.path.exists(target):
            return
        try:
            with open(target, "w") as out:
                out.write(json.dumps(self.generate_graph_json()))
            logger.info("generated graph %s", target)
        except Exception as err:
            logger.error("could not generate graph %s: %s", target, str(err))

    def register(self, ctx: "Context") -> None:
        self._write_graph_file_once()
        for node_name in self.list_node_names():
            ModelNode(self, node_name).register(ctx)


class ModelNode(NodeKind):
    """One node of a ``Model`` behind the ``NodeKind`` interface: every call is forwarded to the model
    with the node's name in place of the request parameters (reference :114-129)."""

    def __init__(self, parent: Model, name: str):
        super().__init__(name)
        self.parent = parent

    def io(self, params: Mapping[str, str]) -> Dict:
        return self.parent.io(self.name)

    def contents(self, params: Mapping[str, str]) -> str:
        return self.parent.contents(self.name)

    def compute(self, params: Mapping[str, str], inputs: Pinout) -> Pinout:
        return self.parent.compute(self.name, inputs)


class Context:
    """Operator registry and per-request executor (reference :132-147)."""

    def __init__(self) -> None:
        self.nodes: Dict[str, NodeKind] = {}

    def register(self, node: NodeKind) -> None:
        key = node.get_name()
        self.nodes[key] = node
        logger.info("Registered node: '%s'", key)

    def get_node(self, name: str) -> NodeKind:
        return self.nodes[name]      # unknown endpoint: KeyError, which /compute reports as HTTP 400

    def compute(self, graph: Graph) -> None:
        """Evaluates the request graph in ``Graph.order()``: each node reads its inputs off its incoming
        edges and leaves its outputs on the outgoing ones."""
        for node in graph.order():
            operator = self.get_node(node.name)
            node.set_pinout(operator.compute(node.params, node.get_pinin()))


_singleton = Context()
_scan_lock = threading.Lock()


def context() -> Context:
    """The process-wide registry (reference :149-152)."""
    return _singleton


def _load_plugin(path: str):
    """Imports one plugin file under its bare stem (as the reference does, :160-168: plugins may import each
    other by that name) and returns the module."""
    stem = os.path.splitext(os.path.basename(path))[0]
    spec = importlib.util.spec_from_file_location(stem, path)
    module = importlib.util.module_from_spec(spec)
    sys.modules[stem] = module
    spec.loader.exec_module(module)
    return module


def scan_nodes(dirs: Iterable[str]) -> None:
    """Plugin discovery (reference :154-176): every ``*.py`` directly under ``<base>/<dir>`` is imported and
    whatever its ``instances()`` returns is registered; a plugin that fails to import or to register is
    logged and skipped."""
    with _scan_lock:
        for subdir in dirs:
            folder = os.path.join(base_dir(), subdir)
            for entry in os.listdir(folder):
                path = os.path.join(folder, entry)
                if not (path.endswith(".py") and os.path.isfile(path)):
                    continue
                try:
                    for plugin in _load_plugin(path).instances():
                        plugin.register(context())
                except Exception as err:
                    logger.info("Could not register '%s': %s", path, str(err))
