"""Host boundary vs golden vectors generated from the REFERENCE's own code
(tests/golden/make_golden.py): Graph.order sequences, wire bytes, error convention, NodeKind /
Model behaviour.  Integer/byte work => bit-exact comparisons."""
import json
import os

import pytest
import torch

from interactive_vit_amd import context as ctxmod
from interactive_vit_amd.context import Context, Model, NodeKind
from interactive_vit_amd.graph import Graph, Pinout
from interactive_vit_amd.message import Request, Response, decode_response, encode_request
from interactive_vit_amd.nodes import cos as cosmod
from interactive_vit_amd.views import compute_bytes, contents, description

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_golden.json")))


def cos_ctx():
    ctx = Context()
    for inst in cosmod.instances():
        inst.register(ctx)
    return ctx


import hashlib


def tensor_record(t):
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


@pytest.mark.parametrize("case", GOLD["order"], ids=lambda c: c["label"])
def test_order_matches_reference(case):
    g = Graph()
    nodes = [g.add_node(f"n{i}", {}) for i in range(case["n"])]
    for a, ach, b, bch in case["edges"]:
        g.connect(nodes[a], ach, nodes[b], bch)
    for b, ch in case["inputs"]:
        g.add_input(torch.zeros(1), nodes[b], ch)
    assert [n.index for n in g.order()] == case["order"]


def test_order_cycle_raises_instead_of_spinning():
    g = Graph()
    a, b = g.add_node("a", {}), g.add_node("b", {})
    g.connect(a, "o", b, "o")
    g.connect(b, "o", a, "o")
    with pytest.raises(Exception, match="cycle"):
        g.order()


def test_order_counter_form_equals_the_plain_rotation_on_random_graphs():
    """Graph.order keeps a readiness counter per node; the schedule must be the reference's rotation (pop the last, emit if all its
    producers are done, else move it to the front) - restated here naively and compared on random DAGs, including multi-edges,
    dangling inputs and nodes listed in any order."""
    import random
    from collections import deque
    from interactive_vit_amd.graph import Graph
    import torch

    def plain(g):
        done, emitted, work = set(), [], deque(g.nodes)
        while work:
            cand = work.pop()
            if Graph._ready(cand, done):
                done.add(cand); emitted.append(cand)
            else:
                work.appendleft(cand)
        return emitted

    rng = random.Random(7)
    for trial in range(200):
        n = rng.randint(1, 14)
        g = Graph()
        nodes = [g.add_node(f"n{i}", {}) for i in range(n)]
        rank = list(range(n)); rng.shuffle(rank)          # a hidden topological rank: edges only go up in it
        for b in range(n):
            for ch in range(rng.randint(0, 3)):
                cands = [a for a in range(n) if rank[a] < rank[b]]
                if cands and rng.random() < 0.8:
                    g.connect(nodes[rng.choice(cands)], f"o{rng.randint(0, 1)}", nodes[b], f"i{ch}")
                else:
                    g.add_input(torch.zeros(1), nodes[b], f"i{ch}")
        assert [x.index for x in g.order()] == [x.index for x in plain(g)], trial


def test_fanout_quirk_preserved():
    g = Graph()
    a, b, c = g.add_node("a", {}), g.add_node("b", {}), g.add_node("c", {})
    e1 = g.connect(a, "o", b, "o")
    e2 = g.connect(a, "o", c, "o")
    a.set_pinout(Pinout({"o": torch.ones(2)}))
    q = GOLD["fanout_quirk"]
    assert (a.outputs["o"] is e2) == q["outputs_o_is_second_edge"]
    assert (e1.tensor is not None) == q["first_consumer_has_tensor"]
    assert (e2.tensor is not None) == q["second_consumer_has_tensor"]


@pytest.mark.parametrize("case", GOLD["wire"], ids=lambda c: c["label"])
def test_wire_bytes_match_reference(case):
    body = bytes.fromhex(case["request_hex"])
    req = Request()
    req.decode(body)
    assert describe_graph(req.graph) == case["decoded"]
    assert str(req.graph) == case["graph_str_before"]
    cos_ctx().compute(req.graph)
    assert str(req.graph) == case["graph_str_after"]
    assert Response(req.graph).encode().hex() == case["response_hex"]
    # and the whole handler in one call
    status, out = compute_bytes(body, cos_ctx())
    assert status == 200 and out.hex() == case["response_hex"]


@pytest.mark.parametrize("case", GOLD["wire"], ids=lambda c: c["label"])
def test_wire_bytes_match_reference_when_written_in_place(case, monkeypatch):
    """Responses of 1 MiB and more are written straight into the bytes object that is returned (message._writable_bytes) instead
    of being joined from pieces; forced here for the reference's golden cases: same bytes, a real immutable `bytes`."""
    from interactive_vit_amd import message
    assert message._writable_bytes(16) is not None, "the in-place path is not available on this interpreter"
    monkeypatch.setattr(message, "_DIRECT_FILL_MIN", 0)
    status, out = compute_bytes(bytes.fromhex(case["request_hex"]), cos_ctx())
    assert status == 200 and type(out) is bytes and out.hex() == case["response_hex"]
    assert hash(out) == hash(bytes.fromhex(case["response_hex"]))


@pytest.mark.parametrize("case", [c for c in GOLD["wire"] if not c["label"].startswith("json_pad_") or c["label"] == "json_pad_0"],
                         ids=lambda c: c["label"])
def test_client_side_codec_round_trip(case):
    """encode_request reproduces the browser-format request bytes; decode_response parses the
    reference's response bytes (net_node.js:56-80, 251-297)."""
    body = bytes.fromhex(case["request_hex"])
    json_size = int.from_bytes(body[12:16], "little")
    spec = json.loads(body[16:16 + json_size].decode())
    req = Request(); req.decode(body)
    tensors = {}
    for e in spec["edges"]:
        if "tensor" in e:
            tensors[e["tensor"]] = req.graph.nodes[e["out_port"]["node"]].inputs[e["out_port"]["channel"]].tensor
    again = encode_request(spec["nodes"], spec["edges"], [tensors[i] for i in sorted(tensors)])
    assert again == body
    blocks = decode_response(bytes.fromhex(case["response_hex"]))
    cos_ctx().compute(req.graph)
    got = [(n.index, ch, t) for n in req.graph.nodes for ch, t in n.get_pinout().pinout.items()]
    assert [(a, b) for a, b, _ in blocks] == [(a, b) for a, b, _ in got]
    for (_, _, t0), (_, _, t1) in zip(blocks, got):
        assert torch.equal(t0, t1)


@pytest.mark.parametrize("case", GOLD["errors"], ids=lambda c: c["label"])
def test_error_convention_matches_reference(case):
    body = bytes.fromhex(case["request_hex"])
    status, out = compute_bytes(body, cos_ctx())
    if case["raises"] is None:
        assert status == 200
        return
    assert status == 400
    if case["raises"] != "AssertionError":   # assertion texts are not part of the contract
        assert out.decode() == case["str"]
    with pytest.raises(Exception) as ei:
        req = Request(); req.decode(body); cos_ctx().compute(req.graph); Response(req.graph).encode()
    assert type(ei.value).__name__ == case["raises"]


def test_cos_metadata():
    c = cosmod.CosNode()
    assert c.get_name() == GOLD["cos"]["name"]
    assert c.io({}) == GOLD["cos"]["io"]
    for params, text in GOLD["cos"]["contents"]:
        assert c.contents(params) == text
    ctx = cos_ctx()
    assert description("cos", {}, ctx) == (200, json.dumps(GOLD["cos"]["io"]).encode())
    assert contents("cos", {"A": "2"}, ctx) == (200, b"cos(2.0x+0.0)")
    assert description("missing", {}, ctx)[0] == 400


def test_nodekind_base_behaviour():
    base = NodeKind("thing")
    assert base.contents({"a": "1", "b": "x y"}) == GOLD["nodekind"]["contents"]
    with pytest.raises(Exception) as e1:
        base.io({})
    assert str(e1.value) == GOLD["nodekind"]["io_raises"]
    with pytest.raises(Exception) as e2:
        base.compute({}, None)
    assert str(e2.value) == GOLD["nodekind"]["compute_raises"]


class Block(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ln = torch.nn.LayerNorm(4)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.GELU(), torch.nn.Linear(8, 4))
        self.scale = torch.nn.Parameter(torch.ones(4))


def test_model_matches_reference(tmp_path):
    gold = GOLD["model_toy"]
    os.makedirs(tmp_path / "static" / "graphs")
    ctxmod.set_base_dir(str(tmp_path))
    toy = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), Block(), torch.nn.Flatten(0))
    toy[0].load_state_dict({k[2:]: torch.tensor(v) for k, v in gold["state_dict"].items()})
    m = Model(toy, "toy")
    assert m.list_node_names() == gold["node_names"]
    assert m.generate_graph_json() == gold["graph_json"]
    ctx = Context()
    m.register(ctx)
    graph_file = tmp_path / "static" / "graphs" / "toy.json"
    assert graph_file.exists() == gold["graph_file_written"]
    assert json.load(open(graph_file)) == gold["graph_json"]
    assert sorted(ctx.nodes.keys()) == gold["registered"]
    for n in gold["node_names"]:
        assert m.contents(n) == gold["contents"][n]
        assert ctx.get_node(n).contents({}) == gold["contents"][n]
    assert m.io(gold["node_names"][0]) == gold["io"]
    g = Graph()
    n0, n1 = g.add_node(gold["node_names"][0], {}), g.add_node(gold["node_names"][1], {})
    g.connect(n0, "o", n1, "o")
    g.add_input(torch.tensor(gold["x"]), n0, "o")
    ctx.compute(g)
    assert torch.allclose(n1.get_pinout().get("o"), torch.tensor(gold["chain_out"]), atol=1e-6)
    # an existing graph file is left alone (reference context.py:100)
    graph_file.write_text("{}")
    Model(toy, "toy").register(Context())
    assert graph_file.read_text() == "{}"


@pytest.mark.parametrize("cnt", sorted(GOLD["model_layout"], key=int))
def test_model_graph_layout(cnt):
    m = Model(torch.nn.Sequential(*[torch.nn.Identity() for _ in range(int(cnt))]), f"seq{cnt}")
    assert [n["pos"] for n in m.generate_graph_json()["nodes"]] == GOLD["model_layout"][cnt]


def test_vgg16_saved_graph_layout_rule():
    """The only Model-generated graph the reference ships: 41 net nodes + 1 category node."""
    v = GOLD["vgg16_graph_shape"]
    n_net = v["n_nodes"] - 1
    m = Model(torch.nn.Sequential(*[torch.nn.Identity() for _ in range(n_net)]), "x")
    gj = m.generate_graph_json()
    assert [n["pos"] for n in gj["nodes"]] == v["pos"][:n_net]
    assert gj["edges"] == v["edges"][:n_net - 1]
    assert v["kinds"] == ["net_node"] * n_net + ["category"]


def test_scan_nodes_discovers_plugins(tmp_path):
    (tmp_path / "main" / "nodes").mkdir(parents=True)
    (tmp_path / "static" / "models").mkdir(parents=True)
    (tmp_path / "static" / "graphs").mkdir(parents=True)
    src = open(cosmod.__file__).read()
    (tmp_path / "main" / "nodes" / "cos.py").write_text(src)
    (tmp_path / "main" / "nodes" / "broken.py").write_text("raise RuntimeError('boom')\n")
    (tmp_path / "main" / "nodes" / "notes.txt").write_text("not python")
    ctxmod.set_base_dir(str(tmp_path))
    before = dict(ctxmod.context().nodes)
    try:
        ctxmod.scan_nodes(["main/nodes", "static/models"])
        assert "cos" in ctxmod.context().nodes       # registered; the broken plugin was skipped
    finally:
        ctxmod.context().nodes.clear()
        ctxmod.context().nodes.update(before)


def test_large_response_direct_fill_equals_join_with_empty_and_scalar_blocks(monkeypatch):
    """the in-place fill of responses >= 1 MiB with an empty tensor and a 0-d tensor among the blocks gives the same
    bytes as the joined form, and a tensor whose size no longer matches its reserved block ends in a clean exception (-> 400)."""
    from interactive_vit_amd import message

    def pin(**kw):
        return Pinout(dict(kw))
    g = Graph()
    for k in range(4):
        g.add_node("n", {})
    big = torch.arange(300000, dtype=torch.float32).reshape(300, 1000)
    g.nodes[0].set_pinout(pin(o=big))
    g.nodes[1].set_pinout(pin(o=torch.zeros((0, 7), dtype=torch.float32)))
    g.nodes[2].set_pinout(pin(o=torch.tensor(3.5)))
    g.nodes[3].set_pinout(pin(o=torch.ones(5)))
    direct = message.Response(g).encode()
    assert len(direct) >= message._DIRECT_FILL_MIN
    monkeypatch.setattr(message, "_DIRECT_FILL_MIN", 1 << 40)
    joined = message.Response(g).encode()
    assert direct == joined
    outs = message.decode_response(direct)
    assert [tuple(t.shape) for _, _, t in outs] == [(300, 1000), (0, 7), (), (5,)]

    class Lying(torch.Tensor):                       # reports one shape when the slot is reserved, carries another
        pass
    monkeypatch.setattr(message, "_DIRECT_FILL_MIN", 1 << 20)
    t = torch.Tensor._make_subclass(Lying, torch.zeros(300000))
    t._ivit_shape = (300001,)
    g.nodes[0].set_pinout(pin(o=t))
    with pytest.raises(Exception, match="changed size"):
        message.Response(g).encode()
