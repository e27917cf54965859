"""Pins the ORACLE itself (CPU only):

* its arithmetic against torch's own modules - nn.Conv2d, nn.MultiheadAttention, nn.LayerNorm,
  nn.GELU, nn.Linear - the building blocks of torchvision's VisionTransformer (independent code path);
* its integer bookkeeping against the plain-C restatement (oracle/patch_index.c) and F.unfold;
* its node-graph results against golden activations produced by driving it through the REFERENCE's
  real Request.decode -> Context.compute -> Response.encode (tests/golden/vit_tiny_golden.json).
"""
import ctypes
import hashlib
import json
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from interactive_vit_amd.context import Context
from interactive_vit_amd.context import Model
from interactive_vit_amd.graph import Pinout
from interactive_vit_amd.message import decode_response, encode_request
from interactive_vit_amd.models.vit import make_vit_model_class
from interactive_vit_amd.views import compute_bytes
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.vit_config import test_config as small_config
from interactive_vit_amd.weights import init_weights, state_digest, synthetic_images
from oracle import vit_oracle as vo
from oracle.cpu_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "vit_tiny_golden.json")))


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libpatch_oracle.so"))
    lib.oracle_unfold_index.restype = ctypes.c_int64
    lib.oracle_unfold_index.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.oracle_unfold_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.oracle_token_row.restype = ctypes.c_int64
    lib.oracle_token_row.argtypes = [ctypes.c_int64, ctypes.c_int32]
    return lib


@pytest.mark.parametrize("image,patch", [(64, 16), (224, 16), (224, 14), (32, 8), (28, 7)])
def test_unfold_index_three_ways(c_oracle, image, patch):
    idx = vo.unfold_index(image, patch)
    g = image // patch
    cidx = np.empty((g * g, 3 * patch * patch), dtype=np.int64)
    assert c_oracle.oracle_unfold_index(image, patch, cidx.ctypes.data) == cidx.size
    assert np.array_equal(idx, cidx)
    # every pixel of the image is used exactly once
    assert np.array_equal(np.sort(idx.reshape(-1)), np.arange(3 * image * image))
    # torch's own unfold (the im2col inside nn.Conv2d) orders columns the same way
    x = torch.arange(3 * image * image, dtype=torch.float32).reshape(1, 3, image, image)
    ref = torch.nn.functional.unfold(x, kernel_size=patch, stride=patch).transpose(1, 2)[0]
    assert torch.equal(ref.to(torch.int64), torch.from_numpy(idx))
    xs = torch.rand(2, 3, image, image)
    cout = np.empty((2, g * g, 3 * patch * patch), dtype=np.float32)
    c_oracle.oracle_unfold_f32(xs.numpy().ctypes.data, cout.ctypes.data, 2, image, patch)
    assert torch.equal(vo.unfold(xs, image, patch), torch.from_numpy(cout))


def test_token_row_remap(c_oracle):
    for patches in (16, 196, 256):
        for m in (0, 1, patches - 1, patches, 3 * patches + 7):
            b, n = divmod(m, patches)
            assert c_oracle.oracle_token_row(m, patches) == vo.token_row(b, 1 + n, patches + 1)


def test_nodes_against_torch_modules():
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    x = synthetic_images(2, cfg, seed=5)
    with torch.no_grad():
        t = vo.transform(x)
        conv = torch.nn.Conv2d(3, cfg.dim, cfg.patch, stride=cfg.patch)
        conv.load_state_dict({"weight": sd["conv_proj.weight"], "bias": sd["conv_proj.bias"]})
        ref = conv(t).flatten(2).transpose(1, 2)
        got = vo.conv_proj(t, sd, cfg)
        assert torch.allclose(got, ref, atol=2e-5)
        tok = vo.tokens(got, sd, cfg)
        assert torch.equal(tok[:, 0], (sd["class_token"][0] + sd["encoder.pos_embedding"][0, :1]).expand(2, -1))
        assert torch.equal(tok[:, 1:], got + sd["encoder.pos_embedding"][:, 1:])

        pre = vo.layer_prefix(0)
        ln1 = torch.nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        ln1.load_state_dict({"weight": sd[pre + "ln_1.weight"], "bias": sd[pre + "ln_1.bias"]})
        ln2 = torch.nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        ln2.load_state_dict({"weight": sd[pre + "ln_2.weight"], "bias": sd[pre + "ln_2.bias"]})
        mha = torch.nn.MultiheadAttention(cfg.dim, cfg.heads, batch_first=True)
        mha.load_state_dict({k: sd[pre + "self_attention." + k] for k in
                             ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias")})
        fc1 = torch.nn.Linear(cfg.dim, cfg.mlp); fc1.load_state_dict({"weight": sd[pre + "mlp.0.weight"], "bias": sd[pre + "mlp.0.bias"]})
        fc2 = torch.nn.Linear(cfg.mlp, cfg.dim); fc2.load_state_dict({"weight": sd[pre + "mlp.3.weight"], "bias": sd[pre + "mlp.3.bias"]})
        h = ln1(tok)
        y = tok + mha(h, h, h, need_weights=False)[0]
        y = y + fc2(torch.nn.GELU()(fc1(ln2(y))))
        assert torch.allclose(vo.encoder_layer(tok, sd, 0, cfg), y, atol=2e-5)

        lnf = torch.nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        lnf.load_state_dict({"weight": sd["encoder.ln.weight"], "bias": sd["encoder.ln.bias"]})
        assert torch.allclose(vo.encoder_ln(y, sd, cfg), lnf(y), atol=2e-5)
        head = torch.nn.Linear(cfg.dim, cfg.classes)
        head.load_state_dict({"weight": sd["heads.head.weight"], "bias": sd["heads.head.bias"]})
        assert torch.allclose(vo.heads(vo.cls(lnf(y)), sd), head(lnf(y)[:, 0]), atol=2e-5)


def test_attention_probabilities_are_a_softmax():
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    h = torch.randn(2, cfg.tokens, cfg.dim)
    out, p = vo.attention(h, sd, 0, cfg, return_probs=True)
    assert torch.allclose(p.sum(-1), torch.ones_like(p.sum(-1)), atol=1e-6)
    out_e, p_e = vo.attention(h.double(), sd, 0, cfg, return_probs=True, emulate=True)
    assert torch.allclose(p_e.sum(-1), torch.ones_like(p_e.sum(-1)), atol=1e-12)
    assert float((out_e - out.double()).abs().max() / out.abs().max()) < 1e-2   # bf16 rounding scale


def test_emulate_mode_is_plain_mode_plus_bf16_rounding():
    """float64 + emulate differs from float64 plain by the bf16 rounding scale, and the weights
    really are bf16-representable in that mode."""
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    x = synthetic_images(2, cfg, seed=5).double()
    plain = vo.forward(x, sd, cfg)["logits"]
    emu = vo.forward(x, sd, cfg, emulate=True)["logits"]
    err = float((plain - emu).abs().max() / plain.abs().max())
    assert 1e-4 < err < 2e-2
    w = vo._w(sd, "heads.head.weight", torch.float64, True)
    assert torch.equal(w, w.to(torch.bfloat16).to(torch.float64))


@pytest.mark.parametrize("ln_fold", [False, True])
def test_emulate_mode_with_every_rounding_point_off_is_the_plain_forward(ln_fold):
    """the rounding-aware mode is what every per-node GPU gate compares against, so it must be "the plain forward + roundings" and
    nothing else.  With every rounding point switched off (ROUND_ONLY = empty set) the emulate path - LayerNorm fold algebra, the engine's softmax
    forms, fused patch / token arithmetic - must reproduce the plain float64 forward to round-off, node by node."""
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    x = synthetic_images(2, cfg, seed=5).double()
    plain = vo.forward(x, sd, cfg, keep=True)
    order = vo.node_suffixes(cfg)
    saved = (vo.ROUND_ONLY, vo.LN_FOLD)
    try:
        vo.ROUND_ONLY, vo.LN_FOLD = frozenset(), ln_fold
        chain = vo.forward(x, sd, cfg, keep=True, emulate=True)
        for i, node in enumerate(order):
            node_in = x if i == 0 else plain[order[i - 1]]
            emu = vo.run_node(node, node_in, sd, cfg, emulate=True)
            err = float((plain[node] - emu).abs().max() / plain[node].abs().max())
            assert err <= 1e-12, (node, err)
    finally:
        vo.ROUND_ONLY, vo.LN_FOLD = saved
    # the whole chain: the emulate forward evaluates the transform in float32, as the engine does (3 flops per element) - the only difference left
    err = float((plain["logits"] - chain["logits"]).abs().max() / plain["logits"].abs().max())
    assert err <= 1e-6, err


def test_centred_fold_algebra_is_the_plain_layer():
    """Round 5: the engine's folded GEMMs multiply rn16(x - m) and add W' m back (include/ivit.h: ivit_ln_fold_calibrate); the oracle mirrors the
    vectors through LN_CENTRE.  With every rounding point off, ANY centre vectors must leave an encoder layer where the plain forward puts it -
    to f32 round-off of the one subtraction the engine makes in f32 - and with the roundings on, centring must not move the layer further from
    the plain forward than the plain copy does when the rows carry a large common offset."""
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    x = synthetic_images(2, cfg, seed=5).double()
    tok = vo.forward(x, sd, cfg, keep=True)["tokens"]
    g = torch.Generator().manual_seed(0)
    off = torch.randn(cfg.dim, generator=g, dtype=torch.float64) * 2.0            # a channel-constant offset on every row
    rows = tok + off
    plain = vo.encoder_layer(rows, sd, 0, cfg)
    saved = (vo.ROUND_ONLY, vo.LN_FOLD, vo.LN_CENTRE)
    try:
        vo.LN_FOLD = True
        vo.LN_CENTRE = torch.randn(2 * cfg.layers, cfg.dim, generator=g) * 3.0   # arbitrary vectors
        vo.ROUND_ONLY = frozenset()
        emu = vo.encoder_layer(rows, sd, 0, cfg, emulate=True)
        assert float((plain - emu).abs().max() / plain.abs().max()) <= 1e-5
        vo.ROUND_ONLY = None
        vo.LN_CENTRE = None
        e_plain_copy = float((plain - vo.encoder_layer(rows, sd, 0, cfg, emulate=True)).abs().max() / plain.abs().max())
        m = rows.reshape(-1, cfg.dim).mean(0).float()                             # what the calibration takes: the per-channel means
        vo.LN_CENTRE = torch.stack([m] * (2 * cfg.layers))
        e_centred = float((plain - vo.encoder_layer(rows, sd, 0, cfg, emulate=True)).abs().max() / plain.abs().max())
        assert e_centred < e_plain_copy, (e_centred, e_plain_copy)
    finally:
        vo.ROUND_ONLY, vo.LN_FOLD, vo.LN_CENTRE = saved


def test_realistic_statistics_weights_have_the_statistics_they_claim():
    """weights.realistic_statistics_weights (what the LayerNorm-fold calibration is tested and benchmarked on: tests/test_gpu_configs.py, bench.py --weights
    realistic): deterministic, the same tensor names and shapes as the seeded set, token rows whose |mean| / std reaches 2 and more (the plain 16-bit copy of such rows is
    what rounds 3-4's guard refused), and offsets that are constant across rows - so the per-channel means the calibration takes remove them."""
    from interactive_vit_amd.weights import realistic_statistics_weights
    cfg = small_config(name="vit_realstats", image=64, patch=16, dim=768, heads=12, layers=1, mlp=3072, classes=8)   # ViT-B width: the hot channels are 4 of 768
    a, b = realistic_statistics_weights(cfg, seed=21), realistic_statistics_weights(cfg, seed=21)
    ref = init_weights(cfg, seed=21, mode="rich")
    assert sorted(a) == sorted(ref) and all(a[k].shape == ref[k].shape and torch.equal(a[k], b[k]) for k in a)
    x = synthetic_images(2, cfg, seed=5)
    tok = vo.forward(x, a, cfg, keep=True)["tokens"].reshape(-1, cfg.dim)
    assert float((tok.mean(-1).abs() / tok.std(-1)).max()) >= 2.0            # the guard statistic is a maximum over rows
    centred = tok - tok.mean(0, keepdim=True)
    ratio = (centred.pow(2).mean(-1) / tok.var(-1, unbiased=False)).sqrt()       # rms of the centred row over the row's own spread: the fold's noise factor
    assert float(ratio.max()) <= 1.12, float(ratio.max())                      # <= 12 % more noise than LayerNorm kernels: the guard's threshold 0.5 as a factor


def test_seeded_weights_and_image_are_pinned():
    cfg = VARIANTS[VGOLD["config"]]
    sd = init_weights(cfg, seed=VGOLD["weights"]["seed"], mode=VGOLD["weights"]["mode"])
    assert state_digest(sd) == VGOLD["weights"]["sha256"]
    img = synthetic_images(1, cfg, seed=VGOLD["image"]["seed"])[0]
    assert hashlib.sha256(img.numpy().tobytes()).hexdigest() == VGOLD["image"]["sha256"]
    assert sum(v.numel() for v in sd.values()) == cfg.param_count()


def test_vit_tiny_node_graph_matches_reference_run():
    """BASELINE config 1: ViT-Ti/16, one 224x224 image, CPU forward through the node path - here
    through THIS repo's Request/Context/Response, compared with the recorded run through the
    reference's (same oracle backend, so only the plumbing differs: results must agree to f32
    round-off and the response framing exactly)."""
    cfg = VARIANTS[VGOLD["config"]]
    sd = init_weights(cfg, seed=0, mode="rich")
    VitModel = make_vit_model_class(Model, Pinout)
    vit = VitModel(cfg, OracleBackend(cfg, sd))
    # (the fixture was recorded through the reference's Context in round 2; the two-channel `.with_attn` layer nodes joined in round 4)
    assert vit.list_node_names() == VGOLD["node_names"] + vit.with_attn_node_names()
    gj = vit.generate_graph_json()
    assert len(gj["nodes"]) == VGOLD["graph_json_nodes"]
    assert [n["pos"] for n in gj["nodes"]] == VGOLD["graph_json_pos"]
    ctx = Context()
    for name in vit.list_node_names():
        from interactive_vit_amd.context import ModelNode
        ModelNode(vit, name).register(ctx)
    chain = vit.chain_node_names()
    img = synthetic_images(1, cfg, seed=1234)[0]
    nodes = [{"endpoint": n, "params": {}} for n in chain]
    edges = [{"tensor": 0, "out_port": {"node": 0, "channel": "o"}}] + [
        {"in_port": {"node": i, "channel": "o"}, "out_port": {"node": i + 1, "channel": "o"}} for i in range(len(chain) - 1)]
    status, resp = compute_bytes(encode_request(nodes, edges, [img]), ctx)
    assert status == 200
    assert len(resp) == VGOLD["response"]["byte_size"]
    blocks = decode_response(resp)
    assert [{"node": a, "channel": b} for a, b, _ in blocks] == VGOLD["response"]["json"]
    for (node, ch, t), rec in zip(blocks, VGOLD["per_node"]):
        assert list(t.shape) == rec["shape"], rec["endpoint"]
        flat = t.flatten()
        samples = flat[::rec["sample_stride"]][:97]
        assert torch.allclose(samples, torch.tensor(rec["samples"]), rtol=1e-4, atol=1e-5 * max(1.0, rec["max_abs"])), rec["endpoint"]
        assert abs(float(flat.abs().max()) - rec["max_abs"]) <= 1e-4 * max(1.0, rec["max_abs"])
    assert torch.allclose(blocks[-1][2], torch.tensor(VGOLD["logits"]), rtol=1e-4, atol=1e-5)


def test_preprocess_oracle_is_the_torchvision_preset():
    """`preprocess` = resize shorter side (antialiased bilinear) -> centre crop S -> normalise, the
    ImageClassification preset of the reference's model plugin (vgg16.py:40-42).  torchvision is not
    installed here, so its size / offset arithmetic is pinned by known cases and the resize by properties."""
    from interactive_vit_amd.vit_config import VARIANTS
    from oracle import vit_oracle
    cfg = VARIANTS["vit_b_16"]
    assert vit_oracle.preprocess_resize(cfg) == 256 and vit_oracle.preprocess_resize(VARIANTS["vit_l_16_384"]) == 439
    g = torch.Generator().manual_seed(5)
    # a constant image stays constant through the (normalised-weight) resize: output = (c - mean) / std
    x = torch.full((1, 3, 300, 500), 0.25)
    y = vit_oracle.preprocess(x, cfg)
    assert y.shape == (1, 3, 224, 224)
    ref = vit_oracle.transform(torch.full((1, 3, 224, 224), 0.25))
    assert torch.allclose(y, ref, atol=1e-6)
    # already 256 on the shorter side and square: no resampling at all, the crop offset is (256 - 224) / 2 = 16
    x = torch.rand((2, 3, 256, 256), generator=g)
    y = vit_oracle.preprocess(x, cfg)
    assert torch.allclose(y, vit_oracle.transform(x[:, :, 16:240, 16:240]), atol=1e-6)
    # portrait and landscape go through the same code with the roles of H and W swapped
    x = torch.rand((1, 3, 333, 517), generator=g)
    a = vit_oracle.preprocess(x, cfg)
    b = vit_oracle.preprocess(x.transpose(2, 3).contiguous(), cfg).transpose(2, 3)
    assert torch.allclose(a, b, atol=1e-6)
    # unbatched form through the node dispatcher
    assert torch.equal(vit_oracle.run_node_any("preprocess", x[0], None, cfg), a[0])


def test_layernorm_fold_is_an_equally_good_bf16_evaluation():
    """The engine's default on the bf16 path folds each LayerNorm into the GEMM that consumes it
    (oracle: vit_oracle.folded_linear / LN_FOLD).  In exact arithmetic the folded form IS the LayerNorm followed by the
    linear layer; with the engine's bf16 rounding points it is as far from the f32 forward as the unfolded form."""
    from interactive_vit_amd.vit_config import test_config
    from oracle import vit_oracle as vo
    cfg = test_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    g = torch.Generator().manual_seed(9)
    x = torch.randn((2, cfg.tokens, cfg.dim), generator=g, dtype=torch.float64) * 1.5 + 0.2
    pre = vo.layer_prefix(0)
    w, b = sd[pre + "mlp.0.weight"].double(), sd[pre + "mlp.0.bias"].double()
    gamma, beta = sd[pre + "ln_2.weight"].double(), sd[pre + "ln_2.bias"].double()
    exact = vo.layer_norm(x, gamma, beta, cfg.ln_eps) @ w.t() + b
    # algebra: without any rounding the fold reproduces LayerNorm + linear to f64 round-off
    mu = x.mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + cfg.ln_eps)
    wf = w * gamma[None, :]
    assert torch.allclose(rstd * (x @ wf.t() - mu * wf.sum(1)) + (w @ beta + b), exact, rtol=1e-10, atol=1e-10)
    # with the engine's rounding points: same order of error as rounding LayerNorm's output
    folded = vo.folded_linear(x, w, b, gamma, beta, cfg.ln_eps)
    unfolded = vo.rnd(vo.layer_norm(x, gamma, beta, cfg.ln_eps), True) @ vo.rnd(w, True).t() + b
    den = exact.abs().max()
    e_fold, e_unf = float((folded - exact).abs().max() / den), float((unfolded - exact).abs().max() / den)
    assert e_fold <= 5e-3 and e_unf <= 5e-3 and e_fold <= 2.0 * e_unf + 1e-4, (e_fold, e_unf)
    # and a whole layer in either mode stays inside the per-node bound against the plain forward
    ref = vo.encoder_layer(x, sd, 0, cfg)
    for fold in (False, True):
        vo.LN_FOLD = fold
        err = float((vo.encoder_layer(x, sd, 0, cfg, emulate=True) - ref).abs().max() / ref.abs().max())
        assert err <= 5e-3, (fold, err)
