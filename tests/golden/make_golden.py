#!/usr/bin/env python3
"""Generates tests/golden/*.json by RUNNING THE REFERENCE in this container.

The reference (0Marble/interactive-vit, read-only at /root/reference) has no tests and no golden
vectors of its own (main/tests.py:1-3), so the vectors that pin this repo's restatement of the
node-graph boundary are produced here from the reference's real code:

* ``main/graph.py`` and ``main/message.py`` import as they are (they need only torch);
* ``main/context.py`` reads one attribute, ``django.conf.settings.BASE_DIR`` (context.py:4,99,156);
  Django is not installed here, so that single attribute is provided by an in-memory module object
  pointing at an empty temporary tree - nothing of Django's behaviour is imitated;
* ``main/nodes/cos.py`` is loaded from where it lies.

Only DATA is written (inputs, expected outputs, wire bytes as hex) - no reference source text.
The GPU box never sees /root/reference; tests read the committed JSON.

Run:  python tests/golden/make_golden.py      (rewrites tests/golden/reference_golden.json and
                                                tests/golden/vit_tiny_golden.json)
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import struct
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def import_reference():
    tmp = tempfile.mkdtemp(prefix="ivit_ref_base_")
    for d in ("main/nodes", "static/models", "static/graphs"):
        os.makedirs(os.path.join(tmp, d))
    conf = types.ModuleType("django.conf")
    conf.settings = types.SimpleNamespace(BASE_DIR=tmp)
    dj = types.ModuleType("django")
    dj.conf = conf
    sys.modules["django"] = dj
    sys.modules["django.conf"] = conf
    import main.graph as rgraph
    import main.message as rmessage
    import main.context as rcontext
    spec = importlib.util.spec_from_file_location("cos", os.path.join(REF, "main/nodes/cos.py"))
    rcos = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rcos)
    return tmp, rgraph, rmessage, rcontext, rcos


def pack_request(nodes, edges, tensors, json_pad_spaces=0):
    """Independent request encoder following nodes/net_node.js:56-175 (not this repo's codec)."""
    js = json.dumps({"nodes": nodes, "edges": edges}) + " " * json_pad_spaces
    jb = js.encode()
    out = bytearray(struct.pack("<4I", 0, 0x69BABE69, len(tensors), len(jb)))
    out += jb
    while len(out) % 4:
        out.append(0)
    for t in tensors:
        a = np.ascontiguousarray(t.numpy(), dtype="<f4")
        dims = list(t.shape)
        out += struct.pack("<2I", 8 + 4 * len(dims) + 4 * a.size, len(dims))
        out += struct.pack(f"<{len(dims)}I", *dims)
        out += a.tobytes()
    struct.pack_into("<I", out, 0, len(out))
    return bytes(out)


def tensor_record(t: torch.Tensor):
    return {"shape": list(t.shape), "sha256": hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest(),
            "values": t.flatten()[:8].tolist()}


def describe_graph(g):
    nodes = []
    for n in g.nodes:
        ins = {}
        for ch, e in n.inputs.items():
            ins[ch] = {"src": None if e.input is None else [e.input.node.index, e.input.channel],
                       "tensor": None if e.tensor is None else tensor_record(e.tensor)}
        outs = {ch: (None if e.output is None else [e.output.node.index, e.output.channel]) for ch, e in n.outputs.items()}
        nodes.append({"name": n.name, "params": n.params, "index": n.index, "inputs": ins, "outputs": outs})
    return nodes


def main():
    tmp, rgraph, rmessage, rcontext, rcos = import_reference()
    gold = {"generated_by": "tests/golden/make_golden.py", "reference": "0Marble/interactive-vit @ /root/reference",
            "torch": torch.__version__}

    # ------------------------------------------------------------------ Graph.order()
    order_cases = []

    def order_case(label, n, edges, inputs=()):
        g = rgraph.Graph()
        nodes = [g.add_node(f"n{i}", {}) for i in range(n)]
        for (a, ach, b, bch) in edges:
            g.connect(nodes[a], ach, nodes[b], bch)
        for (b, ch) in inputs:
            g.add_input(torch.zeros(1), nodes[b], ch)
        order_cases.append({"label": label, "n": n, "edges": edges, "inputs": list(inputs),
                            "order": [x.index for x in g.order()]})

    order_case("chain6", 6, [[i, "o", i + 1, "o"] for i in range(5)], [[0, "o"]])
    order_case("diamond_plus_isolated", 5, [[0, "o", 1, "o"], [0, "p", 2, "o"], [1, "o", 3, "a"], [2, "o", 3, "b"]])
    order_case("reversed_chain", 5, [[i + 1, "o", i, "o"] for i in range(4)])
    order_case("two_chains_interleaved", 6, [[0, "o", 2, "o"], [2, "o", 4, "o"], [1, "o", 3, "o"], [3, "o", 5, "o"]])
    order_case("fan_in_three", 4, [[0, "o", 3, "a"], [1, "o", 3, "b"], [2, "o", 3, "c"]])
    order_case("single", 1, [])
    order_case("empty", 0, [])
    order_case("all_isolated", 4, [])
    order_case("deep_then_wide", 7, [[6, "o", 0, "o"], [0, "o", 1, "o"], [0, "q", 2, "o"], [5, "o", 4, "o"], [4, "o", 3, "o"]])
    gold["order"] = order_cases

    # fan-out quirk: second consumer of one channel replaces the first
    g = rgraph.Graph()
    a, b, c = g.add_node("a", {}), g.add_node("b", {}), g.add_node("c", {})
    e1 = g.connect(a, "o", b, "o")
    e2 = g.connect(a, "o", c, "o")
    p = rgraph.Pinout(); p.set("o", torch.ones(2))
    a.set_pinout(p)
    gold["fanout_quirk"] = {"outputs_o_is_second_edge": a.outputs["o"] is e2, "first_consumer_has_tensor": e1.tensor is not None,
                            "second_consumer_has_tensor": e2.tensor is not None}

    # ------------------------------------------------------------------ wire format + cos through Context.compute
    ctx = rcontext.Context()
    for inst in rcos.instances():
        inst.register(ctx)

    torch.manual_seed(0)
    wire = []

    def wire_case(label, nodes, edges, tensors, pad_spaces=0, compute=True):
        req_bytes = pack_request(nodes, edges, tensors, pad_spaces)
        req = rmessage.Request()
        req.decode(req_bytes)
        rec = {"label": label, "request_hex": req_bytes.hex(), "decoded": describe_graph(req.graph),
               "graph_str_before": str(req.graph)}
        if compute:
            ctx.compute(req.graph)
            rec["graph_str_after"] = str(req.graph)
            rec["response_hex"] = rmessage.Response(req.graph).encode().hex()
        wire.append(rec)

    def cosn(params):
        return {"endpoint": "cos", "params": params}

    def e_t(i, node, ch="o"):
        return {"tensor": i, "out_port": {"node": node, "channel": ch}}

    def e_c(a, b, ach="o", bch="o"):
        return {"in_port": {"node": a, "channel": ach}, "out_port": {"node": b, "channel": bch}}

    x23 = torch.arange(6, dtype=torch.float32).reshape(2, 3) / 7.0
    wire_case("cos_chain_2x3", [cosn({}), cosn({})], [e_t(0, 0), e_c(0, 1)], [x23])
    for pad in range(4):   # json_size % 4 = every residue
        wire_case(f"json_pad_{pad}", [cosn({"A": "2.5", "b": 1})], [e_t(0, 0)], [torch.randn(5)], pad_spaces=pad)
    wire_case("scalar_0d", [cosn({"b": 0.5})], [e_t(0, 0)], [torch.tensor(1.25)])
    wire_case("empty_tensor", [cosn({})], [e_t(0, 0)], [torch.zeros(0, 4)])
    wire_case("rank4", [cosn({"A": -1})], [e_t(0, 0)], [torch.randn(2, 1, 3, 2)])
    wire_case("two_blocks_two_chains", [cosn({}), cosn({"A": 3}), cosn({}), cosn({"b": "2"})],
              [e_t(1, 2), e_t(0, 0), e_c(0, 1), e_c(2, 3)], [torch.randn(4), torch.randn(3, 3)])
    wire_case("out_of_order_nodes", [cosn({"A": 2}), cosn({}), cosn({"b": 1})],
              [e_c(2, 1), e_c(1, 0), e_t(0, 2)], [torch.randn(2, 2)])
    wire_case("no_nodes", [], [], [], compute=True)
    gold["wire"] = wire

    # error convention: what the reference raises (views.compute would answer 400 with str(e))
    errors = []

    def err_case(label, nodes, edges, tensors):
        req_bytes = pack_request(nodes, edges, tensors)
        try:
            req = rmessage.Request(); req.decode(req_bytes); ctx.compute(req.graph)
            rmessage.Response(req.graph).encode()
            errors.append({"label": label, "request_hex": req_bytes.hex(), "raises": None})
        except Exception as ex:  # noqa
            errors.append({"label": label, "request_hex": req_bytes.hex(), "raises": type(ex).__name__, "str": str(ex)})

    err_case("unknown_endpoint", [{"endpoint": "nope", "params": {}}], [e_t(0, 0)], [torch.zeros(1)])
    err_case("cos_missing_input", [cosn({})], [], [])
    err_case("cos_null_params", [{"endpoint": "cos", "params": None}], [e_t(0, 0)], [torch.zeros(2)])
    err_case("fanout_same_channel", [cosn({}), cosn({}), cosn({})], [e_t(0, 0), e_c(0, 1), e_c(0, 2)], [torch.zeros(2)])
    bad_magic = bytearray(pack_request([cosn({})], [e_t(0, 0)], [torch.zeros(1)])); bad_magic[4] ^= 0xFF
    try:
        rmessage.Request().decode(bytes(bad_magic)); errors.append({"label": "bad_magic", "request_hex": bytes(bad_magic).hex(), "raises": None})
    except Exception as ex:  # noqa
        errors.append({"label": "bad_magic", "request_hex": bytes(bad_magic).hex(), "raises": type(ex).__name__, "str": str(ex)})
    gold["errors"] = errors

    # ------------------------------------------------------------------ CosNode metadata
    cos = rcos.CosNode()
    gold["cos"] = {"name": cos.get_name(),
                   "io": cos.io({}),
                   "contents": [[p, cos.contents(p)] for p in ({}, {"A": "2"}, {"b": 1.5}, {"A": 3, "b": "-0.25"})]}
    base = rcontext.NodeKind("thing")
    gold["nodekind"] = {"contents": base.contents({"a": "1", "b": "x y"})}
    for meth in ("io", "compute"):
        try:
            getattr(base, meth)({}, None) if meth == "compute" else base.io({})
        except Exception as ex:  # noqa
            gold["nodekind"][meth + "_raises"] = str(ex)

    # ------------------------------------------------------------------ Model: leaf enumeration, graph json, compute
    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ln = torch.nn.LayerNorm(4)
            self.mlp = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.GELU(), torch.nn.Linear(8, 4))
            self.scale = torch.nn.Parameter(torch.ones(4))   # not a module: never becomes a node

    torch.manual_seed(1)
    toy = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), Block(), torch.nn.Flatten(0))
    m = rcontext.Model(toy, "toy")
    mctx = rcontext.Context()
    m.register(mctx)
    graph_file = os.path.join(tmp, "static/graphs/toy.json")
    xin = torch.randn(2, 3)
    g = rgraph.Graph()
    names = m.list_node_names()
    gnodes = [g.add_node(n, {}) for n in names[:2]]
    g.connect(gnodes[0], "o", gnodes[1], "o")
    g.add_input(xin, gnodes[0], "o")
    mctx.compute(g)
    gold["model_toy"] = {
        "node_names": names,
        "graph_json": m.generate_graph_json(),
        "graph_file_written": os.path.exists(graph_file),
        "graph_file_equals_generated": json.load(open(graph_file)) == m.generate_graph_json(),
        "registered": sorted(mctx.nodes.keys()),
        "contents": {n: m.contents(n) for n in names},
        "io": m.io(names[0]),
        "state_dict": {k: v.tolist() for k, v in toy.state_dict().items() if k.startswith("0.")},
        "x": xin.tolist(),
        "chain_out": gnodes[1].get_pinout().get("o").tolist(),
    }
    for cnt in (1, 2, 3, 4, 5, 9, 10, 17, 41):   # layout rule: floor(sqrt(n)) columns, 200 px pitch
        mm = rcontext.Model(torch.nn.Sequential(*[torch.nn.Identity() for _ in range(cnt)]), f"seq{cnt}")
        gold.setdefault("model_layout", {})[str(cnt)] = [n["pos"] for n in mm.generate_graph_json()["nodes"]]

    # the one saved graph the reference ships that was produced by Model.generate_graph_json (+ category)
    vgg = json.load(open(os.path.join(REF, "static/graphs/vgg16.json")))
    gold["vgg16_graph_shape"] = {
        "n_nodes": len(vgg["nodes"]), "n_edges": len(vgg["edges"]),
        "kinds": [n["instance"]["kind"] for n in vgg["nodes"]],
        "endpoints": [n["instance"].get("endpoint") for n in vgg["nodes"]],
        "pos": [n["pos"] for n in vgg["nodes"]],
        "edges": vgg["edges"], "n_cats": len(vgg["nodes"][-1]["instance"]["cats"]),
    }

    with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
        json.dump(gold, f, indent=0, sort_keys=True)
    print("wrote reference_golden.json", os.path.getsize(os.path.join(HERE, "reference_golden.json")), "bytes")

    # ------------------------------------------------------------------ ViT-Tiny through the reference's node path
    from interactive_vit_amd.models.vit import make_vit_model_class
    from interactive_vit_amd.vit_config import VARIANTS
    from interactive_vit_amd.weights import init_weights, state_digest, synthetic_images
    from oracle.cpu_backend import OracleBackend

    cfg = VARIANTS["vit_ti_16"]
    sd = init_weights(cfg, seed=0, mode="rich")
    VitModel = make_vit_model_class(rcontext.Model, rgraph.Pinout)
    vit = VitModel(cfg, OracleBackend(cfg, sd), categories=[f"class {i}" for i in range(cfg.classes)])
    vctx = rcontext.Context()
    vit.register(vctx)
    img = synthetic_images(1, cfg, seed=1234)[0]
    chain = vit.chain_node_names()
    nodes = [{"endpoint": n, "params": {}} for n in chain]
    edges = [e_t(0, 0)] + [e_c(i, i + 1) for i in range(len(chain) - 1)]
    req = rmessage.Request()
    req.decode(pack_request(nodes, edges, [img]))
    vctx.compute(req.graph)
    resp = rmessage.Response(req.graph).encode()
    # decode the response independently
    byte_size, magic, block_cnt, json_size = struct.unpack_from("<4I", resp, 0)
    labels = json.loads(resp[16:16 + json_size].decode())
    pos = (16 + json_size + 3) // 4 * 4
    per_node = []
    for i in range(block_cnt):
        bsz, nd = struct.unpack_from("<2I", resp, pos)
        dims = struct.unpack_from(f"<{nd}I", resp, pos + 8)
        numel = int(np.prod(dims)) if nd else 1
        a = np.frombuffer(resp, dtype="<f4", count=numel, offset=pos + 8 + 4 * nd)
        stride = max(1, numel // 97)
        per_node.append({"label": labels[i], "endpoint": chain[labels[i]["node"]], "shape": list(dims),
                         "max_abs": float(np.abs(a).max()), "mean": float(a.mean(dtype=np.float64)),
                         "sample_stride": stride, "samples": a[::stride][:97].tolist()})
        pos += bsz
    assert pos == byte_size
    vgold = {
        "config": "vit_ti_16", "weights": {"seed": 0, "mode": "rich", "sha256": state_digest(sd)},
        "image": {"seed": 1234, "sha256": hashlib.sha256(img.numpy().tobytes()).hexdigest()},
        "node_names": vit.list_node_names(),
        "graph_json_nodes": len(vit.generate_graph_json()["nodes"]),
        "graph_json_pos": [n["pos"] for n in vit.generate_graph_json()["nodes"]],
        "response": {"byte_size": byte_size, "block_cnt": block_cnt, "json": labels},
        "per_node": per_node,
        "logits": per_node[-1] and np.frombuffer(resp, dtype="<f4", count=cfg.classes, offset=byte_size - 4 * cfg.classes).tolist(),
    }
    with open(os.path.join(HERE, "vit_tiny_golden.json"), "w") as f:
        json.dump(vgold, f, indent=0, sort_keys=True)
    print("wrote vit_tiny_golden.json", os.path.getsize(os.path.join(HERE, "vit_tiny_golden.json")), "bytes")


if __name__ == "__main__":
    main()
