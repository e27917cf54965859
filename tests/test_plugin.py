"""The ViT model plugin as seen through the operator API (CPU: oracle backend injected - the
product backend needs the GPU and is covered by tests/test_gpu_*.py)."""
import json
import os

import pytest
import torch

from interactive_vit_amd import context as ctxmod
from interactive_vit_amd.context import Context, Model
from interactive_vit_amd.graph import Graph, Pinout
from interactive_vit_amd.models.vit import VitParameters, default_categories, make_vit_model_class, node_suffixes
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.vit_config import test_config as small_config
from interactive_vit_amd.weights import init_weights, synthetic_images, weight_shapes
from oracle import vit_oracle as vo
from oracle.cpu_backend import OracleBackend


@pytest.fixture()
def plugin(tmp_path):
    (tmp_path / "static" / "graphs").mkdir(parents=True)
    ctxmod.set_base_dir(str(tmp_path))
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    vit = make_vit_model_class(Model, Pinout)(cfg, OracleBackend(cfg, sd))
    ctx = Context()
    vit.register(ctx)
    return cfg, sd, vit, ctx, tmp_path


def test_registration_and_graph_file(plugin):
    cfg, sd, vit, ctx, base = plugin
    names = vit.list_node_names()
    assert names == ([f"{cfg.name}:{s}" for s in node_suffixes(cfg)] + [f"{cfg.name}:forward", f"{cfg.name}:preprocess"]
                     + [f"{cfg.name}:encoder.layers.{i}.attn" for i in range(cfg.layers)]
                     + [f"{cfg.name}:encoder.layers.{i}.with_attn" for i in range(cfg.layers)])
    assert sorted(ctx.nodes) == sorted(names)
    assert all("/" not in n for n in names)             # node names are URL path segments (urls.py:12-13)
    gj = json.load(open(base / "static" / "graphs" / f"{cfg.name}.json"))
    chain = vit.chain_node_names()
    assert [n["instance"].get("endpoint") for n in gj["nodes"][:-1]] == chain
    assert gj["nodes"][-1]["instance"]["kind"] == "category"
    assert len(gj["nodes"][-1]["instance"]["cats"]) == cfg.classes
    assert len(gj["edges"]) == len(chain)               # chain edges + the edge into `category`
    assert gj["edges"][-1] == {"in_port": {"node": len(chain) - 1, "channel": "o"}, "out_port": {"node": len(chain), "channel": "o"}}
    for n in names:
        outs = ["o", "attn"] if n.endswith(".with_attn") else ["o"]       # the two-channel layer nodes
        assert ctx.get_node(n).io({}) == {"ins": ["o"], "outs": outs}
        assert ctx.get_node(n).contents({}).startswith(f"<p>{n}</p>")


def test_chain_through_context_matches_direct_forward(plugin):
    cfg, sd, vit, ctx, _ = plugin
    img = synthetic_images(1, cfg, seed=9)[0]
    g = Graph()
    nodes = [g.add_node(n, {}) for n in vit.chain_node_names()]
    for a, b in zip(nodes, nodes[1:]):
        g.connect(a, "o", b, "o")
    g.add_input(img, nodes[0], "o")
    ctx.compute(g)
    logits = nodes[-1].get_pinout().get("o")
    assert logits.shape == (cfg.classes,)                # what the client `category` node expects
    ref = vo.forward(img.unsqueeze(0), sd, cfg)["logits"][0]
    assert torch.allclose(logits, ref, atol=1e-5)
    fwd = ctx.get_node(f"{cfg.name}:forward").compute({}, Pinout({"o": img})).get("o")
    assert torch.allclose(fwd, ref, atol=1e-5)
    # shapes the browser viewers rely on
    assert nodes[0].get_pinout().get("o").shape == (3, cfg.image, cfg.image)
    assert nodes[2].get_pinout().get("o").shape == (cfg.tokens, cfg.dim)


def test_attention_map_node_shapes(plugin):
    cfg, sd, vit, ctx, _ = plugin
    img = synthetic_images(1, cfg, seed=9)[0]
    g = Graph()
    chain = [f"{cfg.name}:transform", f"{cfg.name}:conv_proj", f"{cfg.name}:tokens", f"{cfg.name}:encoder.layers.0",
             f"{cfg.name}:encoder.layers.1.attn"]
    nodes = [g.add_node(n, {}) for n in chain]
    for a, b in zip(nodes, nodes[1:]):
        g.connect(a, "o", b, "o")
    g.add_input(img, nodes[0], "o")
    ctx.compute(g)
    amap = nodes[-1].get_pinout().get("o")
    assert amap.shape == (cfg.heads, cfg.tokens, cfg.tokens)          # [C,H,W]: what MultiView displays
    assert torch.allclose(amap.sum(-1), torch.ones(cfg.heads, cfg.tokens), atol=1e-5)
    assert ctx.get_node(chain[-1]).contents({}).startswith(f"<p>{chain[-1]}</p>")


def test_layer_node_with_attention_channel_through_the_byte_path(plugin):
    """SURVEY 8(f) row 4 as written: the attention map as an EXTRA OUTPUT CHANNEL of the layer node.  `encoder.layers.<i>.with_attn`
    continues the chain through "o" and carries "attn" ([heads,N,N], a [C,H,W] tensor for MultiView, multi_view.js:53-66) beside it;
    Response ships every channel of every node (main/message.py:80-83), so the response JSON lists both."""
    from interactive_vit_amd.message import decode_response, encode_request
    from interactive_vit_amd.views import compute_bytes
    cfg, sd, vit, ctx, _ = plugin
    img = synthetic_images(1, cfg, seed=9)[0]
    p = cfg.name + ":"
    chain = [p + "transform", p + "conv_proj", p + "tokens", p + "encoder.layers.0.with_attn", p + "encoder.layers.1"]
    nodes = [{"endpoint": e, "params": {}} for e in chain]
    edges = ([{"tensor": 0, "out_port": {"node": 0, "channel": "o"}}]
             + [{"in_port": {"node": i, "channel": "o"}, "out_port": {"node": i + 1, "channel": "o"}} for i in range(len(chain) - 1)])
    status, body = compute_bytes(encode_request(nodes, edges, [img]), ctx)
    assert status == 200
    blocks = decode_response(body)
    assert [{"node": a, "channel": c} for a, c, _ in blocks] == [
        {"node": 0, "channel": "o"}, {"node": 1, "channel": "o"}, {"node": 2, "channel": "o"},
        {"node": 3, "channel": "o"}, {"node": 3, "channel": "attn"}, {"node": 4, "channel": "o"}]      # the golden response JSON: both channels of node 3
    by = {(a, c): t for a, c, t in blocks}
    assert by[(3, "attn")].shape == (cfg.heads, cfg.tokens, cfg.tokens) and by[(3, "o")].shape == (cfg.tokens, cfg.dim)
    acts = vo.forward(img.unsqueeze(0), sd, cfg, keep=True)
    assert torch.allclose(by[(3, "o")], acts["encoder.layers.0"][0], atol=1e-5)          # "o" is the plain layer node's output
    assert torch.allclose(by[(4, "o")], acts["encoder.layers.1"][0], atol=1e-5)          # and the chain goes on through it
    assert torch.allclose(by[(3, "attn")], vo.attention_map(acts["tokens"], sd, 0, cfg)[0], atol=1e-6)
    assert torch.allclose(by[(3, "attn")].sum(-1), torch.ones(cfg.heads, cfg.tokens), atol=1e-5)


def test_missing_input_and_unknown_node(plugin):
    cfg, sd, vit, ctx, _ = plugin
    with pytest.raises(AssertionError):
        ctx.get_node(f"{cfg.name}:tokens").compute({}, Pinout())
    with pytest.raises(KeyError):
        ctx.get_node(f"{cfg.name}:encoder.layers.99")
    with pytest.raises(KeyError):
        vit.compute(f"{cfg.name}:nope", Pinout({"o": torch.zeros(1)}))


def test_parameter_container_has_torchvision_names():
    cfg = small_config()
    sd = init_weights(cfg, seed=0)
    mod = VitParameters(sd)
    assert sorted(mod.state_dict().keys()) == sorted(sd.keys()) == sorted(weight_shapes(cfg).keys())
    for k, v in mod.state_dict().items():
        assert torch.equal(v, sd[k])
    with pytest.raises(RuntimeError):
        mod(torch.zeros(1))


def test_categories_from_file(tmp_path):
    p = tmp_path / "cats.txt"
    p.write_text("\n".join(f"label {i}" for i in range(7)))
    assert default_categories(7, str(p))[3] == "label 3"
    assert default_categories(5, str(p)) == [f"class {i}" for i in range(5)]   # wrong length -> placeholders


def test_variant_table_matches_survey():
    assert VARIANTS["vit_b_16"].macs_per_image() == 17563828224
    assert round(VARIANTS["vit_ti_16"].macs_per_image() / 1e9, 3) == 1.254
    assert round(VARIANTS["vit_l_16_384"].macs_per_image() / 1e9, 3) == 191.066
    assert round(VARIANTS["vit_h_14"].macs_per_image() / 1e9, 3) == 167.295
    assert VARIANTS["vit_l_16_384"].tokens == 577 and VARIANTS["vit_h_14"].tokens == 257


def test_pending_tensor_waits_once_and_only_when_read():
    """engine.PendingTensor (the lazily synchronised output of the asynchronous host path, include/ivit.h:
    ivit_forward_host_async): the engine-side bookkeeping must not wait, any torch operation must wait exactly once and
    return plain tensors - checked here against a stand-in engine that counts ivit_host_wait calls (no GPU needed)."""
    import ctypes
    from interactive_vit_amd.engine import PendingTensor

    class Lib:
        def __init__(self):
            self.calls = []

        def ivit_host_wait(self, h, ticket):
            self.calls.append(int(ticket.value))
            return 0

    class Eng:
        def __init__(self):
            self.lib = Lib()
            self._h = ctypes.c_void_p(1)

        def _check(self, rc):
            assert rc == 0

        def _retire(self, ticket):                 # the engine drops the buffers of completed tickets (Engine._hold / _retire)
            self.retired = getattr(self, "retired", []) + [ticket]

    eng = Eng()
    base = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    t = PendingTensor.wrap(base, eng, 7)
    assert isinstance(t, torch.Tensor) and t._ivit_shape == (2, 3) and t._ivit_ptr == base.data_ptr()
    with torch._C.DisableTorchFunctionSubclass():          # what Engine.forward does with a chained input
        _ = t._version
        view = t.detach().to(torch.float32).contiguous()
        assert view.data_ptr() == base.data_ptr()
    assert eng.lib.calls == []
    assert t.numpy().shape == (2, 3)                        # Response.encode's access (reference main/message.py:114)
    assert eng.lib.calls == [7]
    u = t * 2 + t
    assert type(u) is torch.Tensor and torch.equal(u, base * 3) and eng.lib.calls == [7]
    assert torch.equal(t, base) and tuple(t.shape) == (2, 3) and eng.lib.calls == [7]
    t2 = PendingTensor.wrap(base.clone(), eng, 9)
    t2.mul_(0.5)                                            # an in-place write waits first, then bumps the version
    assert eng.lib.calls == [7, 9]


def test_checkpoint_file_in_torchvision_names_loads(tmp_path):
    """SURVEY section 5 "checkpoint / resume" (the reference plugin loads real weights, static/models/vgg16.py:12-14): a local
    safetensors file with torchvision key names - bf16 / f16 / f32 tensors, a flattened conv weight, extra keys - becomes the
    f32 state dict the engine takes; a torch.save file works too; missing tensors and wrong shapes name the offender."""
    from safetensors.torch import save_file
    from interactive_vit_amd.weights import load_state_dict_file, weight_shapes
    cfg = small_config()
    sd = init_weights(cfg, seed=11, mode="rich")
    stored = {}
    for i, (k, v) in enumerate(sd.items()):
        stored[k] = v.to([torch.float32, torch.bfloat16, torch.float16][i % 3]).contiguous()
    stored["conv_proj.weight"] = sd["conv_proj.weight"].reshape(cfg.dim, -1).contiguous()     # flattened [D, 3 p p]
    stored["some.optimizer.state"] = torch.zeros(3)
    path = str(tmp_path / "vit.safetensors")
    save_file(stored, path)
    got = load_state_dict_file(path, cfg)
    assert list(got) == list(weight_shapes(cfg))
    for i, (k, v) in enumerate(sd.items()):
        assert got[k].dtype == torch.float32 and got[k].is_contiguous() and tuple(got[k].shape) == tuple(v.shape)
        want = v if k == "conv_proj.weight" else v.to([torch.float32, torch.bfloat16, torch.float16][i % 3]).to(torch.float32)
        assert torch.equal(got[k], want), k
    # the oracle runs on it (so does the engine: test_gpu_parity.test_checkpoint_file_through_the_engine)
    from oracle import vit_oracle as vo
    x = synthetic_images(1, cfg, seed=1)
    assert torch.isfinite(vo.forward(x, got, cfg)["logits"]).all()
    # torch.save form
    p2 = str(tmp_path / "vit.pt")
    torch.save({"state_dict": sd}, p2)
    got2 = load_state_dict_file(p2, cfg)
    assert all(torch.equal(got2[k], sd[k]) for k in sd)
    # errors name the tensor
    broken = dict(stored); del broken["heads.head.bias"]
    save_file(broken, path)
    with pytest.raises(KeyError, match="heads.head.bias"):
        load_state_dict_file(path, cfg)
    broken = dict(stored); broken["encoder.ln.weight"] = torch.zeros(cfg.dim + 1)
    save_file(broken, path)
    with pytest.raises(ValueError, match="encoder.ln.weight"):
        load_state_dict_file(path, cfg)
    # same element count, wrong layout: a transposed matrix is refused by name, not silently re-viewed
    for key in ("encoder.layers.encoder_layer_0.mlp.0.weight", "encoder.layers.encoder_layer_1.self_attention.in_proj_weight"):
        broken = dict(stored); broken[key] = stored[key].t().contiguous()
        save_file(broken, path)
        with pytest.raises(ValueError, match=key.replace(".", r"\.")):
            load_state_dict_file(path, cfg)
    # the documented position-embedding / class-token layouts are accepted
    ok = dict(stored); ok["encoder.pos_embedding"] = stored["encoder.pos_embedding"].reshape(cfg.tokens, cfg.dim).contiguous()
    ok["class_token"] = stored["class_token"].reshape(cfg.dim).contiguous()
    save_file(ok, path)
    got3 = load_state_dict_file(path, cfg)
    assert tuple(got3["encoder.pos_embedding"].shape) == (1, cfg.tokens, cfg.dim) and tuple(got3["class_token"].shape) == (1, 1, cfg.dim)


def test_category_labels_come_from_a_local_file(tmp_path, monkeypatch):
    """The graph JSON closes with a `category` node carrying the class names (static/models/vgg16.py:16-29); without network the labels
    come from a local file (IVIT_CATEGORIES), one per line."""
    from interactive_vit_amd.models.vit import default_categories
    cfg = small_config()
    f = tmp_path / "labels.txt"
    f.write_text("\n".join(f"label {i}" for i in range(cfg.classes)) + "\n")
    monkeypatch.setenv("IVIT_CATEGORIES", str(f))
    assert default_categories(cfg.classes) == [f"label {i}" for i in range(cfg.classes)]
    f.write_text("too\nfew\n")
    assert default_categories(cfg.classes)[0] == "class 0"          # a file of the wrong length is not used
