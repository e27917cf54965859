"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bars (BASELINE.json north_star / SURVEY 8(d)):

* patch index / unfold bookkeeping: BIT-EXACT;
* floating point: REL_TOL = 1e-3 on max|gpu - ref| / max|ref| (the tolerance north_star states for
  bf16), per node and for the whole forward, against the oracle evaluated with the engine's bf16
  rounding points (``emulate=True``, float64 accumulation) - the only legitimate differences left
  are f32 accumulation order and exp2/erf ulps, so a kernel bug cannot hide inside the tolerance;
* against the PLAIN f32 forward ("the CPU node-graph forward") the distance is the bf16 operand
  rounding itself (~1.6e-3 rms per dot product, oracle/vit_oracle.py header): asserted against the
  looser, documented BF16_VS_F32 bounds and printed, never silently widened.
"""
import numpy as np
import pytest
import torch

from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.vit_config import test_config as small_config
from interactive_vit_amd.weights import init_weights, synthetic_images

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3          # the tolerance north_star states for bf16 (vs the rounding-aware oracle)
# Against the PLAIN f32 forward the distance is the bf16 operand rounding itself.  Bounds = measured on MI355X + 25 %
# (a 3x allowance hides regressions): one node 2.2e-3...2.8e-3 -> 3.5e-3; whole chains per model below.
BF16_VS_F32_NODE = 3.5e-3
E2E_MEASURED = {"vit_test": 4.1e-3, "vit_test80": 3.0e-3, "vit_ti_16": 4.0e-3, "vit_b_16": 9.3e-3, "vit_l_16_384": 4.0e-3, "vit_h_14": 1.0e-2}   # max over the tests' seeds


def e2e_bound(cfg) -> float:
    """Whole-chain bound vs the plain f32 oracle for a model: measured + 25 %."""
    return 1.25 * E2E_MEASURED[cfg.name]


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def strict_nodes(eng, cfg, sd, acts, x, suffixes, tol=REL_TOL):
    """Per-node gate at REL_TOL: each named node alone, fed the oracle's input for it, against the
    oracle with the engine's bf16 rounding points."""
    from oracle import vit_oracle
    order = vit_oracle.node_suffixes(cfg)
    for suffix in suffixes:
        i = order.index(suffix)
        node_in = x if i == 0 else acts[order[i - 1]]
        got = eng.run_node(suffix, node_in.cuda()).cpu()
        emu = vit_oracle.run_node(suffix, node_in.double(), sd, cfg, emulate=True)
        err = rel_err(got, emu)
        print(f"{cfg.name}:{suffix} alone vs rounding-aware oracle {err:.2e}")
        assert err <= tol, f"{cfg.name}:{suffix}: {err:.3e}"


@pytest.fixture(scope="module")
def small():
    from interactive_vit_amd.engine import Engine
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=5)
    yield cfg, sd, eng
    eng.close()


def test_oracle_mirrors_the_engine_rounding_points(small):
    """conftest sets the oracle's LayerNorm-fold switch from IVIT_FOLD_LN; the engine reports what it does."""
    from oracle import vit_oracle
    cfg, sd, eng = small
    assert eng.ln_fold == vit_oracle.LN_FOLD


def test_library_is_the_hip_build(built_lib):
    from interactive_vit_amd import engine
    lib = engine.load_library()
    assert lib.ivit_abi_version() == engine.ABI_VERSION
    assert b"gfx950" in lib.ivit_build_info()


def test_unfold_bit_exact(small):
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(3, cfg, seed=11).cuda()
    # no transform: pure gather + bf16 rounding
    got = eng.debug_unfold(x, normalise=False).cpu()
    ref = vit_oracle.unfold(x.cpu(), cfg.image, cfg.patch).reshape(-1, cfg.patch_k).to(torch.bfloat16).float()
    assert torch.equal(got, ref)
    # fused transform + unfold: (x-mean)/std in f32, then the same rounding
    got = eng.debug_unfold(x, normalise=True).cpu()
    ref = vit_oracle.unfold(vit_oracle.transform(x.cpu()), cfg.image, cfg.patch).reshape(-1, cfg.patch_k)
    assert torch.equal(got, ref.to(torch.bfloat16).float())


def test_unfold_index_identity_image(small):
    """Every pixel carries its own flat offset (exactly representable: offsets < 2^24 in f32 and
    the test image is small enough for bf16 only through the index map) -> compare index maps."""
    from oracle import vit_oracle
    from interactive_vit_amd.engine import unfold_offset
    cfg, _, _ = small
    idx = vit_oracle.unfold_index(cfg.image, cfg.patch)
    for n in (0, 1, cfg.grid, cfg.patches - 1):
        for k in (0, 1, cfg.patch, cfg.patch * cfg.patch, cfg.patch_k - 1):
            assert unfold_offset(cfg.image, cfg.patch, n, k) == idx[n, k]


@pytest.mark.parametrize("batch", [1, 3])
def test_every_node_matches_oracle(small, batch):
    """Each node alone, fed the ORACLE's input for that node (errors do not accumulate)."""
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(batch, cfg, seed=5)
    acts = vit_oracle.forward(x, sd, cfg, keep=True)
    cur = x
    for suffix in vit_oracle.node_suffixes(cfg):
        ref = acts[suffix]
        got = eng.run_node(suffix, cur.cuda()).cpu()
        assert got.shape == ref.shape, suffix
        emu = vit_oracle.run_node(suffix, cur.double(), sd, cfg, emulate=True)
        err = rel_err(got, emu)
        assert err <= REL_TOL, f"{suffix}: rel err vs rounding-aware oracle {err:.3e}"
        err32 = rel_err(got, ref)
        assert err32 <= BF16_VS_F32_NODE, f"{suffix}: rel err vs plain f32 oracle {err32:.3e}"
        cur = ref


def test_unbatched_interactive_shapes(small):
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(1, cfg, seed=6)[0]            # [3,S,S] as the browser sends it
    cur = x
    for suffix in vit_oracle.node_suffixes(cfg):
        ref = vit_oracle.run_node_any(suffix, cur, sd, cfg)
        emu = vit_oracle.run_node_any(suffix, cur.double(), sd, cfg, emulate=True)
        got = eng.run_node(suffix, cur)                # CPU tensor in -> host path -> CPU tensor out
        assert got.device.type == "cpu" and got.dtype == torch.float32
        assert got.shape == ref.shape
        assert rel_err(got, emu) <= REL_TOL, suffix
        cur = ref


def test_fused_forward_matches_chain_and_oracle(small):
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(5, cfg, seed=7)
    ref = vit_oracle.forward(x, sd, cfg, keep=True)
    emu = vit_oracle.forward(x.double(), sd, cfg, keep=True, emulate=True)
    logits, cls = eng.forward(x.cuda(), 0, len(eng.stages), want_cls=True)
    # whole-chain comparisons sit at the bf16 rounding floor against EITHER oracle: a difference d
    # in front of a rounding to a grid of spacing u comes out as ~sqrt(d*u), so chained roundings
    # decorrelate the two computations up to the rounding noise itself (per-node gate: test above)
    assert rel_err(logits, emu["logits"]) <= e2e_bound(cfg)
    assert rel_err(logits, ref["logits"]) <= e2e_bound(cfg)
    # class-token features after encoder.ln feed `heads`; cls_out is the [B,D] f32 row of it
    assert rel_err(cls, emu["cls"]) <= e2e_bound(cfg)
    assert rel_err(cls, ref["cls"]) <= e2e_bound(cfg)
    # node-by-node on the GPU gives the SAME bits as the fused range (same kernels, same order)
    cur = x.cuda()
    for suffix in vit_oracle.node_suffixes(cfg):
        cur = eng.run_node(suffix, cur)
    assert torch.equal(cur.cpu(), logits.cpu())


def test_stage_ranges_compose(small):
    cfg, sd, eng = small
    x = synthetic_images(2, cfg, seed=8).cuda()
    full = eng.forward(x, 0, len(eng.stages))
    ns = len(eng.stages)
    for cut in (1, 2, 3, 4, ns - 3, ns - 2, ns - 1):
        a = eng.forward(x, 0, cut)
        b = eng.forward(a, cut, ns)
        assert torch.equal(b, full), f"cut at {cut}"


def test_fused_range_takes_layer0_statistics_from_the_patch_gemm(small):
    """A range that runs the patch embedding INTO an encoder layer lets the patch GEMM's epilogue (EPI_BIAS_ROWADD_STATS, the `_rs` kernel)
    leave the LayerNorm statistics pairs and the 16-bit copy of the token rows; ivit_row_stats_pairs then visits the class rows only.  Same
    bytes as the range cut in front of the layer, where the statistics kernel reads the whole token stream."""
    cfg, sd, eng = small
    if not eng.ln_fold_for(3):
        pytest.skip("LayerNorm fold off")
    x = synthetic_images(3, cfg, seed=41).cuda()

    def profiled(fn):
        eng.profile(True); eng.profile_reset()
        out = fn()
        kern = eng.profile_kernels(); eng.profile(False)
        return out, kern

    fused, kern = profiled(lambda: eng.forward(x, 0, 4))                  # transform .. encoder.layers.0
    patch = [k for k in kern if k.startswith("patch:")]
    assert len(patch) == 1 and patch[0].endswith("_rs"), sorted(kern)
    assert sum(v["launches"] for k, v in kern.items() if k.startswith("layernorm")) == 1      # the class rows
    tok, kern_t = profiled(lambda: eng.forward(x, 0, 3))                  # ends on the tokens: plain row-add epilogue
    assert not [k for k in kern_t if k.startswith("patch:")][0].endswith("_rs"), sorted(kern_t)
    assert torch.equal(eng.forward(tok, 3, 4), fused)


def test_errors_are_exceptions_with_messages(small):
    from interactive_vit_amd.engine import EngineError
    cfg, sd, eng = small
    with pytest.raises(EngineError, match="expects input"):
        eng.run_node("tokens", torch.zeros(3, 3))
    with pytest.raises(EngineError, match="batch"):
        eng.forward(torch.zeros(6, 3, cfg.image, cfg.image), 0, 1)     # > max_batch
    with pytest.raises(EngineError, match="stage range"):
        eng.forward(torch.zeros(cfg.patches, cfg.dim), 2, 2)


def test_engine_on_an_absent_device_is_refused_with_a_message(small):
    """ivit_config.device selects the GPU; kernel attributes are kept per (device, kernel) inside the library
    (csrc/kernels_gemm.hip: ensure_dynamic_lds), and a device this process cannot see is an error, not device 0."""
    from interactive_vit_amd.engine import Engine, EngineError
    cfg, sd, eng = small
    n = torch.cuda.device_count()
    with pytest.raises(EngineError, match="not present"):
        Engine(cfg, sd, device=n, max_batch=1)
    # engines created and destroyed repeatedly on the same device keep working (the attribute cache is per device, not per engine)
    x = synthetic_images(1, cfg, seed=3).cuda()
    ref = eng.forward(x, 0, len(eng.stages))
    for _ in range(2):
        e2 = Engine(cfg, sd, device=0, max_batch=1)
        try:
            assert torch.equal(e2.forward(x, 0, len(e2.stages)), ref)
        finally:
            e2.close()


def test_vit_tiny_forward():
    """BASELINE config 1 model (ViT-Ti/16, 197 tokens) on one image and a ragged batch."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_ti_16"]
    sd = init_weights(cfg, seed=0, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=3)
    try:
        x = synthetic_images(3, cfg, seed=1234)
        acts = vit_oracle.forward(x, sd, cfg, keep=True)
        emu = vit_oracle.forward(x.double(), sd, cfg, keep=True, emulate=True)
        logits = eng.forward(x.cuda(), 0, len(eng.stages)).cpu()
        e_emu, e_f32 = rel_err(logits, emu["logits"]), rel_err(logits, acts["logits"])
        print(f"vit_ti_16 logits (whole chain): vs rounding-aware oracle {e_emu:.2e}, vs plain f32 {e_f32:.2e}")
        assert e_emu <= e2e_bound(cfg)
        assert e_f32 <= e2e_bound(cfg)
        mid = eng.forward(x.cuda(), 0, 3 + 6).cpu()          # after encoder layer 5
        assert rel_err(mid, acts["encoder.layers.5"]) <= e2e_bound(cfg)
        strict_nodes(eng, cfg, sd, acts, x, ["conv_proj", "encoder.layers.0", "encoder.layers.11", "heads"])
    finally:
        eng.close()


def test_vit_b16_batch_parity_and_properties():
    """BASELINE config 2 sizes (ViT-B/16): oracle on a 2-image sample (seconds of CPU), plus
    size-independent properties at the full batch of 64: batch independence (image i alone ==
    image i inside the batch, bit for bit) and determinism."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_b_16"]
    sd = init_weights(cfg, seed=0, mode="spec")
    eng = Engine(cfg, sd, device=0, max_batch=64)
    try:
        x = synthetic_images(64, cfg, seed=1234)
        xg = x.cuda()
        logits = eng.forward(xg, 0, len(eng.stages))
        again = eng.forward(xg, 0, len(eng.stages))
        assert torch.equal(logits, again)
        ref = vit_oracle.forward(x[:2], sd, cfg)
        emu = vit_oracle.forward(x[:2].double(), sd, cfg, emulate=True)
        e_emu, e_f32 = rel_err(logits[:2], emu["logits"]), rel_err(logits[:2], ref["logits"])
        print(f"vit_b_16 logits (whole chain): vs rounding-aware oracle {e_emu:.2e}, vs plain f32 {e_f32:.2e}")
        assert e_emu <= e2e_bound(cfg)
        assert e_f32 <= e2e_bound(cfg)
        acts = vit_oracle.forward(x[:2], sd, cfg, keep=True)
        strict_nodes(eng, cfg, sd, acts, x[:2], ["conv_proj", "encoder.layers.0", "encoder.layers.7", "heads"])
        # 24 images = 4728 token rows: the QKV projection now takes the staggered 256x256 GEMM tile
        # (interactive_vit_amd/csrc/gemm256s_kernel.h); same per-node bar
        acts24 = vit_oracle.forward(x[:24], sd, cfg, keep=True)
        strict_nodes(eng, cfg, sd, acts24, x[:24], ["encoder.layers.0", "encoder.layers.11"])
        for i in (0, 17, 63):
            alone = eng.forward(xg[i:i + 1].contiguous(), 0, len(eng.stages))
            assert torch.equal(alone[0], logits[i]), f"image {i} depends on its batch"
        assert torch.isfinite(logits).all()
    finally:
        eng.close()


@pytest.mark.parametrize("precision,knobs", [("bf16", {}), ("f16", {}), ("f16x", {}), ("f16x", {"IVIT_F16X_MLP2": "1"})])
def test_fused_mlp_is_bit_identical_to_the_two_gemm_launches(precision, knobs, monkeypatch):
    """The fused MLP kernel (csrc/mlp_fused_kernel.h: LN2-fold -> up -> GELU -> down -> residual -> statistics in one launch, weights streamed from a
    packed fragment-native copy) against the two GEMM launches it replaces, through the C ABI: the same engine configuration with IVIT_FUSED_MLP=0 / 1
    must give the same BYTES - a single layer node (no statistics for a next layer), the whole forward (statistics pairs and 16-bit copies chained
    through twelve layers), at the bench batch (197 full workgroups) and at a batch whose last workgroup is ragged.  The dispatched kernels are asserted.
    (F16X: its default split set - hi / lo pairs of the up weight only - and the round-4 set with the down weight split too.)"""
    from interactive_vit_amd.engine import Engine
    cfg = VARIANTS["vit_b_16"]
    sd = init_weights(cfg, seed=0, mode="rich")
    x = synthetic_images(64, cfg, seed=11).cuda()
    outs = {}
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for fused in ("0", "1"):
        monkeypatch.setenv("IVIT_FUSED_MLP", fused)
        eng = Engine(cfg, sd, device=0, max_batch=64, precision=precision)
        try:
            tok = eng.forward(x, 0, 3)                                     # [B, N, D] token stream
            eng.profile(True); eng.profile_reset()
            layer = eng.run_node("encoder.layers.5", tok)
            logits = eng.forward(x, 0, len(eng.stages))
            kern = eng.profile_kernels()
            eng.profile(False)
            has_fused = any(k.startswith("mlp:ivit_mlp_fused_") for k in kern)
            assert has_fused == (fused == "1"), sorted(kern)
            if fused == "1":
                assert not any(k.startswith("mlp1:") or k.startswith("mlp2:") for k in kern), sorted(kern)
            ragged = eng.forward(x[:63].contiguous(), 0, len(eng.stages))   # 12 411 token rows: the last workgroup has 59 of its 64 rows
            one = eng.forward(x[:1].contiguous(), 0, len(eng.stages))       # the interactive path: never fused
            outs[fused] = (layer, logits, ragged, one)
        finally:
            eng.close()
    for a, b, what in zip(outs["0"], outs["1"], ("layer node", "forward", "ragged forward", "one image")):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), f"{precision}: {what} differs between the fused MLP kernel and the two GEMM launches"
    assert torch.equal(outs["1"][3][0], outs["1"][1][0]), "image 0 alone (GEMM pair on small tiles) != image 0 in the batch (fused kernel)"


def test_vit_l16_384_long_sequence():
    """BASELINE config 3 shapes (ViT-L/16 at 384^2: 577 tokens, D = 1024, 16 heads): the attention
    kernel's 16-queries-per-wave / 608-key instantiation, LayerNorm<4>, K = 1024/4096 GEMMs.  The CPU
    oracle is expensive here, so: strict per-node gate on two nodes for one image, plus the
    size-independent properties (determinism, batch independence) on a ragged batch of 3."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_l_16_384"]
    sd = init_weights(cfg, seed=0, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=3)
    try:
        x = synthetic_images(3, cfg, seed=77)
        xg = x.cuda()
        ns = len(eng.stages)
        logits = eng.forward(xg, 0, ns)
        assert torch.isfinite(logits).all()
        assert torch.equal(logits, eng.forward(xg, 0, ns))
        alone = eng.forward(xg[1:2].contiguous(), 0, ns)
        assert torch.equal(alone[0], logits[1])
        # strict gate: tokens -> encoder.layers.0 (attention over 577 keys) and conv_proj (K = 768 GEMM)
        x1 = x[:1]
        t = vit_oracle.transform(x1)
        tok = vit_oracle.tokens(vit_oracle.conv_proj(t, sd, cfg), sd, cfg)
        acts = {"transform": t, "conv_proj": vit_oracle.conv_proj(t, sd, cfg), "tokens": tok}
        strict_nodes(eng, cfg, sd, acts, x1, ["conv_proj", "encoder.layers.0"])
        # first layers of the chain stay inside the bf16 whole-chain bound
        mid_ref = vit_oracle.encoder_layer(vit_oracle.encoder_layer(tok, sd, 0, cfg), sd, 1, cfg)
        mid = eng.forward(x1.cuda(), 0, 5).cpu()
        assert rel_err(mid, mid_ref) <= e2e_bound(cfg)
    finally:
        eng.close()


@pytest.mark.parametrize("image,tokens", [(384, 577), (368, 530), (272, 290)])
@pytest.mark.parametrize("precision", ["bf16", "f16", "f16x", "fp8", "fp8m"])
def test_long_sequence_attention_kernel_every_precision(image, tokens, precision):
    """ivit_attention_q32 (32-query tiled kernel, >= 289 tokens at head dim 64) through every output form it has - 16-bit (bf16 / f16),
    high + low parts for the split out-projection (f16x), e4m3 (fp8) - on a two-head model with ViT-L/16-384's token count (19 query blocks
    on 16 waves: three leftover blocks shared by groups of waves through LDS), with 530 tokens (17 blocks: the one leftover block shared by
    all 16 waves, partials filling LDS to the byte) and 290 (10 blocks, 10 waves, nothing left over).  Each layer node alone against the oracle
    with the engine's rounding points (which evaluates this kernel's arithmetic: integer reference exponent, row sum of rounded numerators)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = small_config(name=f"vit_test_long{tokens}", image=image, patch=16, dim=128, heads=2, layers=2, mlp=256, classes=16)
    assert cfg.tokens == tokens and cfg.head_dim == 64 and vit_oracle.engine_attention_form(cfg.tokens, cfg.head_dim) == "q32"
    sd = init_weights(cfg, seed=11, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=2, precision=precision)
    try:
        vit_oracle.OPERAND_DTYPE = eng.operand_dtype
        vit_oracle.SPLIT_GEMMS = eng.split_gemms
        vit_oracle.LN_FOLD = eng.ln_fold
        x = synthetic_images(2, cfg, seed=17)
        scales = eng.calibrate_fp8(x) if precision in ("fp8", "fp8m") else None
        acts = vit_oracle.forward(x, sd, cfg, keep=True)
        order = vit_oracle.node_suffixes(cfg)
        for i in range(cfg.layers):
            node_in = acts[order[order.index(f"encoder.layers.{i}") - 1]]
            eng.profile(True); eng.profile_reset()
            got = eng.run_node(f"encoder.layers.{i}", node_in.cuda()).cpu()
            kern = eng.profile_kernels()
            eng.profile(False)
            assert "attention:ivit_attention_q32" in kern, sorted(kern)
            if scales:
                emu = vit_oracle.encoder_layer_fp8(node_in.double(), sd, i, cfg, scales[4 * i:4 * i + 4], mlp_only=precision == "fp8m")
                tol = FP8_NODE_TOL
            else:
                emu = vit_oracle.run_node(f"encoder.layers.{i}", node_in.double(), sd, cfg, emulate=True)
                tol = REL_TOL
            err, e32 = rel_err(got, emu), rel_err(got, acts[f"encoder.layers.{i}"])
            print(f"{cfg.name} {precision} encoder.layers.{i} alone vs rounding-aware oracle {err:.2e}, vs plain f32 {e32:.2e}")
            assert err <= tol, (i, err)
            assert torch.isfinite(got).all()
            assert torch.equal(got, eng.run_node(f"encoder.layers.{i}", node_in.cuda()).cpu())
        if precision in ("f16", "f16x"):
            logits = eng.forward(x.cuda(), 0, len(eng.stages)).cpu()
            assert rel_err(logits, acts["logits"]) <= F16_VS_F32_NODE
    finally:
        vit_oracle.OPERAND_DTYPE = torch.bfloat16
        vit_oracle.SPLIT_GEMMS = frozenset()
        eng.close()


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_long_sequence_attention_redo_when_a_late_key_dominates(precision):
    """ivit_attention_q32 takes its reference exponent from the first 32 keys a wave sees and never rescales; a numerator that leaves the
    16-bit (or f32) range shows up as a non-finite row sum and the block is redone with the rows' true maximum.  Here key 100 scores ~ 200
    nats above everything in the first tile for every query (weights built for it): the output must be finite and equal the oracle's."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = small_config(name="vit_test_long290", image=272, patch=16, dim=128, heads=2, layers=1, mlp=256, classes=16)
    sd = init_weights(cfg, seed=12, mode="rich")
    pre = "encoder.layers.encoder_layer_0."
    d, hd, jstar = cfg.dim, cfg.head_dim, 100
    w_in, b_in = sd[pre + "self_attention.in_proj_weight"].clone(), sd[pre + "self_attention.in_proj_bias"].clone()
    w_in[: 2 * d] = 0
    b_in[: 2 * d] = 0
    b_in[:d] = 40.0 / 8.0                      # q = 5 in every component: q . k = 40 sum(k) / 8 ...
    w_in[d: 2 * d, 0] = 4.0 / 8.0              # ... k = h[0] / 2 in every component: score = 40 x 4 x h[0] / 8 = 20 h[0] nats
    sd[pre + "self_attention.in_proj_weight"], sd[pre + "self_attention.in_proj_bias"] = w_in, b_in
    sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"] = torch.ones(d), torch.zeros(d)
    g = torch.Generator().manual_seed(3)
    tok = torch.randn(2, cfg.tokens, d, generator=g)
    tok[:, jstar, 0] += 60.0                   # LayerNorm leaves h[jstar][0] ~ 11: 220 nats against |20 h[0]| <~ 70 elsewhere
    eng = Engine(cfg, sd, device=0, max_batch=2, precision=precision)
    try:
        vit_oracle.OPERAND_DTYPE = eng.operand_dtype
        vit_oracle.SPLIT_GEMMS = eng.split_gemms
        vit_oracle.LN_FOLD = eng.ln_fold
        probs = vit_oracle.attention_map(tok.double(), sd, 0, cfg)
        assert float(probs[:, :, :, jstar].min()) > 0.999     # the construction does what it says
        got = eng.run_node("encoder.layers.0", tok.cuda()).cpu()
        assert torch.isfinite(got).all()
        emu = vit_oracle.run_node("encoder.layers.0", tok.double(), sd, cfg, emulate=True)
        err = rel_err(got, emu)
        print(f"{cfg.name} {precision}: late dominant key, layer vs rounding-aware oracle {err:.2e}")
        assert err <= REL_TOL
    finally:
        vit_oracle.OPERAND_DTYPE = torch.bfloat16
        vit_oracle.SPLIT_GEMMS = frozenset()
        eng.close()


@pytest.mark.parametrize("image,tokens", [(384, 577), (368, 530)])
@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_long_sequence_attention_rows_far_below_zero(image, tokens, precision):
    """ivit_attention_q32 let the keys past N (zero K rows: score exactly 0) through the tile loop and subtracted their numerators from the
    row sum afterwards - a softmax that depended on where 0 lies: for a row whose real logits are all around -32 nats the padded terms are 2^46 each
    (they swamp the f32 sum; in f16 they are infinite and the redo pass, whose maximum counted the padded zeros, then flushed every real numerator
    to 0).  The padded keys are masked now.  Every query's logits here lie within +-2 nats of -32 (q and k mostly bias): the layer must match the
    shift-invariant oracle at 577 tokens (31 padded keys in the last tile) and 530 (14)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = small_config(name=f"vit_test_long{tokens}", image=image, patch=16, dim=128, heads=2, layers=1, mlp=256, classes=16)
    assert cfg.tokens == tokens and vit_oracle.engine_attention_form(cfg.tokens, cfg.head_dim) == "q32"
    sd = init_weights(cfg, seed=13, mode="rich")
    pre = "encoder.layers.encoder_layer_0."
    d = cfg.dim
    w_in, b_in = sd[pre + "self_attention.in_proj_weight"].clone(), sd[pre + "self_attention.in_proj_bias"].clone()
    w_in[: 2 * d] *= 0.25                       # q, k: a little of the input ...
    b_in[:d] = -2.0                             # ... on top of q = -2, k = +2 in every component: q . k / 8 = -4 x 64 / 8 = -32 nats
    b_in[d: 2 * d] = 2.0
    sd[pre + "self_attention.in_proj_weight"], sd[pre + "self_attention.in_proj_bias"] = w_in, b_in
    g = torch.Generator().manual_seed(5)
    tok = torch.randn(2, cfg.tokens, d, generator=g)
    eng = Engine(cfg, sd, device=0, max_batch=2, precision=precision)
    try:
        vit_oracle.OPERAND_DTYPE = eng.operand_dtype
        vit_oracle.SPLIT_GEMMS = eng.split_gemms
        vit_oracle.LN_FOLD = eng.ln_fold
        qkv = vit_oracle.layer_norm(tok.double(), sd[pre + "ln_1.weight"].double(), sd[pre + "ln_1.bias"].double(), cfg.ln_eps) @ w_in.double().t() + b_in.double()
        q, k = qkv[..., :d].reshape(2, -1, cfg.heads, 64).transpose(1, 2), qkv[..., d:2 * d].reshape(2, -1, cfg.heads, 64).transpose(1, 2)
        sc = q @ k.transpose(-1, -2) / 8.0
        assert float(sc.max()) < -20.0, float(sc.max())        # the construction does what it says: every logit far below 0
        eng.profile(True); eng.profile_reset()
        got = eng.run_node("encoder.layers.0", tok.cuda()).cpu()
        kern = eng.profile_kernels(); eng.profile(False)
        assert "attention:ivit_attention_q32" in kern, sorted(kern)
        assert torch.isfinite(got).all()
        emu = vit_oracle.run_node("encoder.layers.0", tok.double(), sd, cfg, emulate=True)
        err = rel_err(got, emu)
        print(f"{cfg.name} {precision}: logits in [{float(sc.min()):.1f}, {float(sc.max()):.1f}] nats, layer vs rounding-aware oracle {err:.2e}")
        assert err <= REL_TOL
    finally:
        vit_oracle.OPERAND_DTYPE = torch.bfloat16
        vit_oracle.SPLIT_GEMMS = frozenset()
        eng.close()


def test_attention_map_nodes(small):
    """`encoder.layers.<i>.attn`: [N,D] -> [heads,N,N] attention probabilities."""
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(3, cfg, seed=12)
    acts = vit_oracle.forward(x, sd, cfg, keep=True)
    for layer in range(cfg.layers):
        node_in = acts["tokens"] if layer == 0 else acts[f"encoder.layers.{layer - 1}"]
        emu = vit_oracle.attention_map(node_in.double(), sd, layer, cfg, emulate=True)
        got = eng.run_node(f"encoder.layers.{layer}.attn", node_in.cuda()).cpu()
        assert got.shape == (3, cfg.heads, cfg.tokens, cfg.tokens)
        assert rel_err(got, emu) <= REL_TOL
        assert torch.allclose(got.sum(-1), torch.ones(3, cfg.heads, cfg.tokens), atol=1e-5)   # rows are distributions
        one = eng.run_node(f"encoder.layers.{layer}.attn", node_in[0])                          # unbatched, host path
        assert one.shape == (cfg.heads, cfg.tokens, cfg.tokens) and one.device.type == "cpu"
        assert torch.equal(one, got[0])


def test_layer_node_attention_channel_equals_the_inspector_bit_for_bit(small):
    """`encoder.layers.<i>.with_attn` (SURVEY 8(f) row 4 as written): channel "o" is the layer node's output, channel "attn" the map
    from the layer's OWN q|k|v - the bytes of the `.attn` inspector node, without its second LayerNorm + QKV GEMM; device and host path."""
    from oracle import vit_oracle
    cfg, sd, eng = small
    x = synthetic_images(3, cfg, seed=12)
    acts = vit_oracle.forward(x, sd, cfg, keep=True)
    for layer in range(cfg.layers):
        node_in = acts["tokens"] if layer == 0 else acts[f"encoder.layers.{layer - 1}"]
        both = eng.run_node_multi(f"encoder.layers.{layer}.with_attn", node_in.cuda())
        assert sorted(both) == ["attn", "o"]
        assert torch.equal(both["o"], eng.run_node(f"encoder.layers.{layer}", node_in.cuda()))
        assert torch.equal(both["attn"], eng.run_node(f"encoder.layers.{layer}.attn", node_in.cuda()))
        assert rel_err(both["attn"], vit_oracle.attention_map(node_in.double(), sd, layer, cfg, emulate=True)) <= REL_TOL
        host = eng.run_node_multi(f"encoder.layers.{layer}.with_attn", node_in[1])            # unbatched CPU tensor: the host path
        assert host["o"].device.type == "cpu" and host["attn"].shape == (cfg.heads, cfg.tokens, cfg.tokens)
        assert torch.equal(host["o"], both["o"][1].cpu()) and torch.equal(host["attn"], both["attn"][1].cpu())


def test_attention_map_197_tokens():
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_ti_16"]
    sd = init_weights(cfg, seed=0, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=2)
    try:
        x = synthetic_images(2, cfg, seed=5)
        tok = vit_oracle.forward(x, sd, cfg, keep=True)["tokens"]
        emu = vit_oracle.attention_map(tok.double(), sd, 0, cfg, emulate=True)
        got = eng.attention_map(0, tok.cuda()).cpu()
        assert got.shape == (2, cfg.heads, 197, 197)
        assert rel_err(got, emu) <= REL_TOL
    finally:
        eng.close()


def test_host_path_graph_replay_is_bit_identical(small):
    """Small-batch requests on the host-buffer entry replay a captured hipGraph from the second
    call on; results must equal the eager first call bit for bit, for several stage ranges, and a
    different input must give a different (correct) output through the same graph."""
    from oracle import vit_oracle
    cfg, sd, eng = small
    ns = len(eng.stages)
    x1 = synthetic_images(1, cfg, seed=31)[0]
    x2 = synthetic_images(1, cfg, seed=32)[0]
    for (b, e_) in ((0, ns), (0, 5), (3, ns)):
        src1 = x1 if b == 0 else vit_oracle.forward(x1.unsqueeze(0), sd, cfg, keep=True)["tokens"][0]
        src2 = x2 if b == 0 else vit_oracle.forward(x2.unsqueeze(0), sd, cfg, keep=True)["tokens"][0]
        first = eng.forward(src1, b, e_)            # eager + capture
        again = eng.forward(src1, b, e_)            # graph replay
        third = eng.forward(src1, b, e_)
        assert torch.equal(first, again) and torch.equal(first, third)
        other = eng.forward(src2, b, e_)            # same graph, new input bytes
        assert not torch.equal(other, first)
        dev = eng.forward(src2.cuda(), b, e_).cpu() # device path never uses graphs
        assert torch.equal(other, dev)


def test_chained_layer_nodes_reuse_the_statistics_the_previous_node_left(small):
    """A node chain hands encoder layer k's output to layer k+1 by reference (host path).  Layer k's MLP-down GEMM then leaves the
    LayerNorm statistics pairs and the 16-bit copy of its output behind (EPI_BIAS_RESID_STATS, as inside a fused range), and layer
    k+1 skips ivit_row_stats_pairs - which writes the very same pairs, so the bytes do not depend on who made them."""
    cfg, sd, eng = small
    if not eng.ln_fold_for(1):
        pytest.skip("LayerNorm fold off")
    img = synthetic_images(1, cfg, seed=31)[0]
    tok = eng.forward(img, 0, 3)                       # transform .. tokens
    indep = [eng.forward(tok.clone(), 3, 4)]
    indep.append(eng.forward(indep[0].clone(), 4, 5))

    def stats_launches(fn):
        eng.profile(True); eng.profile_reset()
        out = fn()
        kern = eng.profile_kernels(); eng.profile(False)
        return out, sum(v["launches"] for k, v in kern.items() if k.startswith("layernorm")), kern

    a, n_a, kern_a = stats_launches(lambda: eng.forward(tok.clone(), 3, 4))
    assert n_a == 1 and any(k.startswith("mlp2:") and k.endswith("_rs") for k in kern_a), sorted(kern_a)   # leaves the pairs
    b, n_b, kern_b = stats_launches(lambda: eng.forward(a, 4, 5))                # same object handed on
    assert n_b == 0, sorted(kern_b)
    assert torch.equal(a, indep[0]) and torch.equal(b, indep[1])
    # anything else that takes the workspaces in between invalidates them: the statistics kernel runs again, same bytes
    a2 = eng.forward(tok.clone(), 3, 4)
    eng.forward(synthetic_images(1, cfg, seed=32)[0], 0, 2)
    b2, n_b2, _ = stats_launches(lambda: eng.forward(a2, 4, 5))
    assert n_b2 == 1 and torch.equal(b2, indep[1])
    # a modified tensor is uploaded again and gets its own statistics
    a3 = eng.forward(tok.clone(), 3, 4)
    a3.mul_(0.5)
    b3, n_b3, _ = stats_launches(lambda: eng.forward(a3, 4, 5))
    assert n_b3 == 1 and torch.equal(b3, eng.forward(a3.clone(), 4, 5))


def test_chained_host_calls_consume_the_resident_copy(small):
    """Context.compute hands node k's output tensor to node k+1 by reference; the
    engine then continues from the device-resident copy instead of uploading it again
    (ivit_forward_host_chained).  The result must be bit-identical to independent calls, and every way
    the residency can be stale must fall back to the bytes of the tensor actually passed."""
    cfg, sd, eng = small
    ns = len(eng.stages)
    img = synthetic_images(1, cfg, seed=21)[0]
    # independent calls: every stage gets a fresh copy of its input (no identity -> plain upload)
    indep, x = [], img
    for s in range(ns):
        x = eng.forward(x.clone(), s, s + 1)
        indep.append(x)
    for rep in range(3):                      # rep 0 eager + capture, later reps replay the graphs
        x = img
        for s in range(ns):
            x = eng.forward(x, s, s + 1)      # same object handed on: chained
            assert torch.equal(x, indep[s]), f"stage {s} differs when chained (rep {rep})"
    # stale residency 1: another request ran in between
    a = eng.forward(img, 0, 1)
    eng.forward(synthetic_images(1, cfg, seed=22)[0], 0, 2)
    assert torch.equal(eng.forward(a, 1, 2), indep[1])
    # stale residency 2: the tensor was modified in place after it was returned
    a = eng.forward(img, 0, 1)
    a.mul_(0.5)
    assert torch.equal(eng.forward(a, 1, 2), eng.forward(a.clone(), 1, 2))
    assert not torch.equal(eng.forward(a.clone(), 1, 2), indep[1])
    # the raw entry point: a foreign token is ignored, a matching one is honoured
    import ctypes
    xin = indep[2].contiguous()
    out1 = torch.empty_like(indep[3]); out2 = torch.empty_like(indep[3])
    tok = ctypes.c_uint64(0)
    assert eng.lib.ivit_forward_host_chained(eng._h, 3, 4, 1, ctypes.c_void_p(xin.data_ptr()), ctypes.c_void_p(out1.data_ptr()),
                                             out1.numel(), ctypes.c_uint64(0xdeadbeef), ctypes.byref(tok)) == 0
    assert tok.value != 0 and torch.equal(out1, indep[3])
    garbage = torch.zeros_like(indep[3])      # with a valid token the host bytes are not read at all
    out3 = torch.empty_like(indep[4])
    assert eng.lib.ivit_forward_host_chained(eng._h, 4, 5, 1, ctypes.c_void_p(garbage.data_ptr()), ctypes.c_void_p(out3.data_ptr()),
                                             out3.numel(), tok, ctypes.byref(tok)) == 0
    assert torch.equal(out3, indep[4])


def test_preprocess_node_matches_oracle(small):
    """`<model>:preprocess`: raw image of any size -> antialiased resize, centre crop,
    normalise.  Oracle = ATen's own antialiased bilinear (torch.nn.functional.interpolate) + torchvision's
    size / offset rules; the kernel sums the two filter dimensions at once, so agreement is to f32 round-off."""
    from oracle import vit_oracle
    from interactive_vit_amd.engine import EngineError
    cfg, sd, eng = small
    s_ = cfg.image
    g = torch.Generator().manual_seed(11)
    for (h, w) in ((s_ * 2 + 5, s_ * 3 + 1), (s_ * 3, s_ * 2), (s_, s_), (s_ + 7, s_ * 8), (s_ * 256 // 224 + 1, s_ * 256 // 224 + 1)):
        x = torch.rand((3, h, w), generator=g)
        ref = vit_oracle.preprocess(x[None].double(), cfg)[0]
        got = eng.run_node("preprocess", x)
        assert got.shape == (3, s_, s_) and got.dtype == torch.float32
        err = float((got.double() - ref).abs().max())
        assert err <= 2e-5, f"{h}x{w}: max abs err {err:.2e}"
        dev = eng.run_node("preprocess", x.cuda()).cpu()
        assert torch.equal(dev, got), "host and device entry points differ"
    # batched, and chained into conv_proj without an upload (the result stays resident like any host output)
    xb = torch.rand((3, 3, s_ * 2, s_ * 2 + 9), generator=g)
    pb = eng.run_node("preprocess", xb)
    assert torch.allclose(pb.double(), vit_oracle.preprocess(xb.double(), cfg), atol=2e-5)
    one = eng.run_node("preprocess", xb[0])
    chained = eng.run_node("conv_proj", one)
    assert torch.equal(chained, eng.run_node("conv_proj", one.clone()))
    with pytest.raises(EngineError):
        eng.run_node("preprocess", torch.rand((4, s_, s_)))


def test_concurrent_compute_from_many_threads(small):
    """Django serves /compute on concurrent threads and the reference takes no locks:
    one engine must give every thread its own correct answer.  Calls are serialised inside the
    library; ctypes releases the GIL around them."""
    import threading
    cfg, sd, eng = small
    ns = len(eng.stages)
    inputs = [synthetic_images(1 + (i % 3), cfg, seed=100 + i) for i in range(8)]
    expected = [eng.forward(x, 0, ns) for x in inputs]                    # sequential, host path
    results = [[None] * 6 for _ in inputs]
    errors = []

    def worker(i):
        try:
            for rep in range(6):
                x = inputs[i] if rep % 2 == 0 else inputs[i].cuda()       # mix host and device entry points
                results[i][rep] = eng.forward(x, 0, ns).cpu()
        except Exception as ex:  # noqa
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, exp in enumerate(expected):
        for got in results[i]:
            assert torch.equal(got, exp), f"thread {i} got another request's result"


def test_head_dim_80_and_patch_14_small():
    """ViT-H/14-shaped small model: head dim 80 (attention's 176-B-row instantiation with a zero-padded
    third MFMA step), patch 14 (K = 588, padded to 640; the unfold kernel's generic per-element path)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = small_config(name="vit_test80", image=56, patch=14, dim=320, heads=4, layers=2, mlp=640, classes=24)
    assert cfg.head_dim == 80 and cfg.patch_k == 588 and cfg.tokens == 17
    sd = init_weights(cfg, seed=5, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=4)
    try:
        x = synthetic_images(4, cfg, seed=9)
        # bookkeeping stays bit-exact on the generic unfold path
        got = eng.debug_unfold(x.cuda(), normalise=True).cpu()
        ref = vit_oracle.unfold(vit_oracle.transform(x), cfg.image, cfg.patch).reshape(-1, cfg.patch_k)
        assert torch.equal(got, ref.to(torch.bfloat16).float())
        acts = vit_oracle.forward(x, sd, cfg, keep=True)
        strict_nodes(eng, cfg, sd, acts, x, vit_oracle.node_suffixes(cfg))
        logits = eng.forward(x.cuda(), 0, len(eng.stages))
        assert rel_err(logits, acts["logits"]) <= e2e_bound(cfg)
        # the fused range (patch GEMM leaving layer 0's statistics: 5 slots of 64 columns, 17 tokens, K padded 588 -> 640) and the node
        # chain (statistics kernel over the whole stream) give the same bits
        cur = x.cuda()
        for suffix in vit_oracle.node_suffixes(cfg):
            cur = eng.run_node(suffix, cur)
        assert torch.equal(cur.cpu(), logits.cpu())
        amap = eng.run_node("encoder.layers.1.attn", acts["encoder.layers.0"].cuda()).cpu()
        emu = vit_oracle.attention_map(acts["encoder.layers.0"].double(), sd, 1, cfg, emulate=True)
        assert rel_err(amap, emu) <= REL_TOL
    finally:
        eng.close()


def test_vit_h14_bf16_shapes():
    """BASELINE config 5's MODEL (ViT-H/14: 257 tokens, D = 1280, 16 heads of 80, MLP 5120, K = 588) in
    bf16 (the fp8 data path of that config has its own tests: tests/test_gpu_configs.py, test_config5_*)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_h_14"]
    sd = init_weights(cfg, seed=0, mode="spec")
    eng = Engine(cfg, sd, device=0, max_batch=2)
    try:
        x = synthetic_images(2, cfg, seed=3)
        ns = len(eng.stages)
        logits = eng.forward(x.cuda(), 0, ns)
        assert torch.isfinite(logits).all() and torch.equal(logits, eng.forward(x.cuda(), 0, ns))
        x1 = x[:1]
        t = vit_oracle.transform(x1)
        acts = {"transform": t, "conv_proj": vit_oracle.conv_proj(t, sd, cfg)}
        acts["tokens"] = vit_oracle.tokens(acts["conv_proj"], sd, cfg)
        strict_nodes(eng, cfg, sd, acts, x1, ["conv_proj"])
        # An encoder layer is itself a chain of five bf16 rounding points (LN out, q|k|v, P, attention
        # out, GELU out); the sqrt(d*u) re-amplification of tiny differences at each of them grows with
        # the reduction lengths (K = 1280 / 5120 here): 5.6e-4 on ViT-B, 6.8e-4 on ViT-L, 1.0e-3 on
        # ViT-H, measured.  GEMM-only nodes stay at 4e-7.  Bar for this config: 2e-3.
        strict_nodes(eng, cfg, sd, acts, x1, ["encoder.layers.0"], tol=2e-3)
        ref = vit_oracle.forward(x1, sd, cfg)["logits"]
        assert rel_err(logits[:1], ref) <= e2e_bound(cfg)
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------
# fp8 data path (BASELINE config 5).  The reference gives no bound for fp8: the gate is
# (a) per layer, fed the oracle's input, against the oracle evaluated with the engine's OWN fp8 policy
# and scales (oracle/vit_oracle.py: encoder_layer_fp8) - e4m3 has a 2^-4 relative grid, a flipped
# rounding of one operand element moves a K-long dot product by ~2^-4 / sqrt(K), and the max over a
# few 1e5 outputs picks the worst of the rare flips: FP8_NODE_TOL on max|d|/max|ref| (measured
# 5e-4..9e-4 on the small model, 1.6e-2 on ViT-H/14) plus FP8_NODE_L2 on |d|_2/|ref|_2;
# (b) the distance to the plain f32 forward, measured and bounded by FP8_VS_F32_E2E.
FP8_NODE_TOL = 3e-2
FP8_NODE_L2 = 5e-3
FP8_VS_F32_E2E = 1.5e-1


def rel_l2(got, ref):
    got = got.detach().double().cpu(); ref = ref.detach().double().cpu()
    return float((got - ref).norm() / ref.norm())


def test_fp8_path_small():
    from interactive_vit_amd.engine import Engine, EngineError
    from oracle import vit_oracle
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=4, precision="fp8")
    try:
        x = synthetic_images(4, cfg, seed=41)
        with pytest.raises(EngineError, match="not calibrated"):
            eng.forward(x.cuda(), 0, len(eng.stages))
        scales = eng.calibrate_fp8(x)
        assert len(scales) == 4 * cfg.layers and all(s > 0 for s in scales)
        ref_scales = vit_oracle.fp8_calibration_scales(x.double(), sd, cfg)
        for a, b in zip(scales, ref_scales):
            assert abs(a - b) <= 1e-2 * b, (scales, ref_scales)     # same amax up to bf16 rounding of the tensors
        acts = vit_oracle.forward(x, sd, cfg, keep=True)
        order = vit_oracle.node_suffixes(cfg)
        for i in range(cfg.layers):
            node_in = acts[order[order.index(f"encoder.layers.{i}") - 1]]
            got = eng.run_node(f"encoder.layers.{i}", node_in.cuda()).cpu()
            emu = vit_oracle.encoder_layer_fp8(node_in.double(), sd, i, cfg, scales[4 * i:4 * i + 4])
            err = rel_err(got, emu)
            print(f"{cfg.name} fp8 encoder.layers.{i} alone vs fp8 oracle {err:.2e} (l2 {rel_l2(got, emu):.2e})")
            assert err <= FP8_NODE_TOL and rel_l2(got, emu) <= FP8_NODE_L2
        # nodes outside the encoder are the bf16 ones
        strict_nodes(eng, cfg, sd, acts, x, ["conv_proj", "heads"])
        logits = eng.forward(x.cuda(), 0, len(eng.stages))
        assert torch.equal(logits, eng.forward(x.cuda(), 0, len(eng.stages)))
        e_f32 = rel_err(logits, acts["logits"])
        e_fp8 = rel_err(logits, vit_oracle.forward_fp8(x.double(), sd, cfg, scales)["logits"])
        print(f"{cfg.name} fp8 logits: vs fp8 oracle {e_fp8:.2e}, vs plain f32 {e_f32:.2e}")
        assert e_f32 <= FP8_VS_F32_E2E and e_fp8 <= FP8_VS_F32_E2E
    finally:
        eng.close()


def test_fp8_vit_h14():
    """BASELINE config 5: ViT-H/14, fp8 weights + activations."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_h_14"]
    sd = init_weights(cfg, seed=0, mode="spec")
    eng = Engine(cfg, sd, device=0, max_batch=4, precision="fp8")
    try:
        x = synthetic_images(4, cfg, seed=3)
        scales = eng.calibrate_fp8(x)
        ns = len(eng.stages)
        logits = eng.forward(x.cuda(), 0, ns)
        assert torch.isfinite(logits).all() and torch.equal(logits, eng.forward(x.cuda(), 0, ns))
        x1 = x[:1]
        tok = vit_oracle.tokens(vit_oracle.conv_proj(vit_oracle.transform(x1), sd, cfg), sd, cfg)
        got = eng.run_node("encoder.layers.0", tok.cuda()).cpu()
        emu = vit_oracle.encoder_layer_fp8(tok.double(), sd, 0, cfg, scales[:4])
        err = rel_err(got, emu)
        print(f"vit_h_14 fp8 encoder.layers.0 alone vs fp8 oracle {err:.2e} (l2 {rel_l2(got, emu):.2e})")
        # A layer chains five quantisations (h8 -> q|k|v bf16 -> att8 -> h8 -> u8); on the coarse e4m3 grid
        # (u = 2^-4) a difference d in front of one re-emerges as ~sqrt(d*u), so two correct evaluations
        # drift apart to ~1 % over the chain at K = 1280/5120 (5e-4 on the small model, same kernels).
        # It stays an order below the fp8-vs-f32 distance of the layer itself, which is the check here.
        plain = vit_oracle.encoder_layer(tok, sd, 0, cfg)
        assert err <= FP8_NODE_TOL and rel_l2(got, emu) <= 2e-2
        assert rel_l2(got, emu) < 0.5 * rel_l2(emu, plain) or rel_l2(got, emu) <= FP8_NODE_L2
        ref = vit_oracle.forward(x1, sd, cfg)["logits"]
        e_f32 = rel_err(logits[:1], ref)
        print(f"vit_h_14 fp8 logits vs plain f32 {e_f32:.2e}")
        assert e_f32 <= FP8_VS_F32_E2E
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------
# f16-operand data path (IVIT_PRECISION_F16): north_star's tolerance against the PLAIN f32 forward.
# gfx950 multiplies f16 at the bf16 rate; 11 significant bits put the operand-rounding noise of a dot
# product at ~2e-4, so here every node is gated at 1e-3 against the CPU f32 node-graph forward itself
# (what the reference's sub(x) returns, main/context.py:79-88) - no rounding-aware oracle needed for
# the claim, though the strict one (mirroring f16 rounding points) is checked too.
F16_VS_F32_NODE = 1e-3


@pytest.mark.parametrize("precision", ["f16", "f16x"])
def test_f16_every_node_within_1e3_of_the_plain_f32_forward(precision):
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=3, precision=precision)
    try:
        assert eng.operand_dtype == torch.float16
        vit_oracle.OPERAND_DTYPE = torch.float16
        vit_oracle.SPLIT_GEMMS = eng.split_gemms
        vit_oracle.LN_FOLD = eng.ln_fold
        x = synthetic_images(3, cfg, seed=5)
        acts = vit_oracle.forward(x, sd, cfg, keep=True)
        cur = x
        for suffix in vit_oracle.node_suffixes(cfg):
            got = eng.run_node(suffix, cur.cuda()).cpu()
            e32 = rel_err(got, acts[suffix])
            emu = rel_err(got, vit_oracle.run_node(suffix, cur.double(), sd, cfg, emulate=True))
            print(f"f16 {cfg.name}:{suffix} vs plain f32 {e32:.2e}, vs f16-rounding oracle {emu:.2e}")
            assert e32 <= F16_VS_F32_NODE, (suffix, e32)
            assert emu <= REL_TOL, (suffix, emu)
            cur = acts[suffix]
        logits = eng.forward(x.cuda(), 0, len(eng.stages)).cpu()
        e = rel_err(logits, acts["logits"])
        print(f"f16 {cfg.name} logits (whole chain) vs plain f32 {e:.2e}")
        assert e <= F16_VS_F32_NODE
        assert torch.equal(logits, eng.forward(x.cuda(), 0, len(eng.stages)).cpu())
    finally:
        vit_oracle.OPERAND_DTYPE = torch.bfloat16
        vit_oracle.SPLIT_GEMMS = frozenset()
        eng.close()


@pytest.mark.parametrize("precision,chain_tol", [("f16", 1.3e-3), ("f16x", 1e-3)])
def test_f16_vit_b16_batch64_nodes_and_chain(precision, chain_tol):
    """BASELINE config 2's shapes on the f16 data paths: per node 1e-3 vs the plain f32 oracle; the whole 12-layer chain (50 GEMMs)
    against the plain f32 forward: IVIT_PRECISION_F16X (the tolerance mode: MLP weights as hi + lo pairs) is gated at north_star's 1e-3
    (7.1e-4 by the oracle's emulation over 3 seeds x 8 images, profiles/r04_f16x_split_sets.txt); IVIT_PRECISION_F16 at its measured 1.0e-3 + 25 %."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_b_16"]
    sd = init_weights(cfg, seed=0, mode="spec")
    eng = Engine(cfg, sd, device=0, max_batch=64, precision=precision)
    try:
        vit_oracle.OPERAND_DTYPE = torch.float16
        vit_oracle.SPLIT_GEMMS = eng.split_gemms
        x = synthetic_images(64, cfg, seed=1234)
        xg = x.cuda()
        logits = eng.forward(xg, 0, len(eng.stages))
        assert torch.equal(logits, eng.forward(xg, 0, len(eng.stages)))
        acts = vit_oracle.forward(x[:2], sd, cfg, keep=True)
        e = rel_err(logits[:2], acts["logits"])
        e4 = rel_err(logits[:4], vit_oracle.forward(x[:4], sd, cfg)["logits"])
        print(f"{precision} vit_b_16 logits (whole chain, B = 64) vs plain f32: {e:.2e} (2 images), {e4:.2e} (4 images)")
        assert max(e, e4) <= chain_tol                     # bf16: 9.3e-3
        order = vit_oracle.node_suffixes(cfg)
        # full batch through single nodes (the tiles the benchmark batch dispatches), two images checked
        tok = vit_oracle.tokens(vit_oracle.conv_proj(vit_oracle.transform(x), sd, cfg), sd, cfg)
        eng.profile(True); eng.profile_reset()
        out0 = eng.run_node("encoder.layers.0", tok.cuda())
        kern = eng.profile_kernels(); eng.profile(False)
        assert any(k.startswith("qkv:ivit_gemm_f16_") for k in kern) and any(k.startswith("mlp:ivit_mlp_fused_f16") for k in kern), sorted(kern)
        vit_oracle.LN_FOLD = eng.ln_fold_for(64)
        e32 = rel_err(out0[:2], acts["encoder.layers.0"])
        emu = rel_err(out0[:2], vit_oracle.run_node("encoder.layers.0", tok[:2].double(), sd, cfg, emulate=True))
        print(f"f16 vit_b_16 encoder.layers.0 at B = 64: vs plain f32 {e32:.2e}, vs f16-rounding oracle {emu:.2e}")
        assert e32 <= F16_VS_F32_NODE and emu <= REL_TOL
        for suffix in ("conv_proj", "encoder.layers.7", "heads"):
            i = order.index(suffix)
            node_in = x[:2] if i == 0 else acts[order[i - 1]]
            got = eng.run_node(suffix, node_in.cuda()).cpu()
            e32 = rel_err(got, acts[suffix])
            print(f"f16 vit_b_16:{suffix} vs plain f32 {e32:.2e}")
            assert e32 <= F16_VS_F32_NODE, (suffix, e32)
        alone = eng.forward(xg[17:18].contiguous(), 0, len(eng.stages))
        assert torch.equal(alone[0], logits[17]), "image 17 depends on its batch"
    finally:
        vit_oracle.OPERAND_DTYPE = torch.bfloat16
        vit_oracle.SPLIT_GEMMS = frozenset()
        eng.close()


@pytest.mark.parametrize("model", ["vit_l_16_384", "vit_h_14"])
def test_tolerance_mode_vit_l_and_h_chain_within_1e3_of_plain_f32(model):
    """the tolerance mode (IVIT_PRECISION_F16X) against the PLAIN f32 oracle - what the reference's sub(x) returns
    (main/context.py:79-88), chained node to node (main/context.py:143-147) - on the 24- and 32-layer models too: one image, the
    whole chain and two single nodes at north_star's 1e-3.  (CPU emulation of the same rounding points, 1 seed: ViT-L/16-384 7.4e-4,
    ViT-H/14 6.3e-4 - profiles/r04_f16x_split_sets.txt.)"""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS[model]
    sd = init_weights(cfg, seed=0, mode="spec")
    eng = Engine(cfg, sd, device=0, max_batch=4, precision="f16x")
    try:
        assert eng.split_gemms >= frozenset({"patch", "head", "proj", "mlp1w"})
        x = synthetic_images(4, cfg, seed=1234)               # four images, not one
        logits = eng.forward(x.cuda(), 0, len(eng.stages)).cpu()
        acts = vit_oracle.forward(x, sd, cfg, keep=True)      # emulate=False: the plain f32 forward
        e = rel_err(logits, acts["logits"])
        per = [rel_err(logits[i:i + 1], acts["logits"][i:i + 1]) for i in range(4)]
        print(f"f16x {model} logits (whole chain, {cfg.layers} layers, 4 images) vs plain f32: {e:.2e}; per image " + ", ".join(f"{v:.2e}" for v in per))
        assert e <= 1e-3 and max(per) <= 1e-3, (e, per)
        order = vit_oracle.node_suffixes(cfg)
        for suffix in ("encoder.layers.0", f"encoder.layers.{cfg.layers - 1}"):
            i = order.index(suffix)
            got = eng.run_node(suffix, acts[order[i - 1]].cuda()).cpu()
            e32 = rel_err(got, acts[suffix])
            print(f"f16x {model}:{suffix} vs plain f32 {e32:.2e}")
            assert e32 <= 1e-3, (suffix, e32)
        # node by node through the chain = the fused range, bit for bit (the reference chains nodes; the bench runs the range)
        cur = x[:1].cuda()
        for suffix in order:
            cur = eng.run_node(suffix, cur)
        assert torch.equal(cur.cpu(), logits[:1])
    finally:
        eng.close()


def test_tolerance_mode_three_weight_seeds_eight_images():
    """VERDICT r3 #1(a) on the ENGINE itself, not only by emulation: IVIT_PRECISION_F16X on ViT-B/16 for three weight seeds x eight images,
    logits against the PLAIN f32 oracle (main/context.py:79-88: what the reference's sub(x) returns), per image and over the batch.  Bound:
    north_star's 1e-3; the measured worst case is printed (round 4: 6.2e-4 per batch of 8, 6.8e-4 on the worst single image = 32 % margin; profiles/r04_f16x_split_sets.txt)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    cfg = VARIANTS["vit_b_16"]
    worst_image = worst_batch = 0.0
    for seed in (0, 1, 2):
        sd = init_weights(cfg, seed=seed, mode="spec")
        x = synthetic_images(8, cfg, seed=1234 + seed)
        eng = Engine(cfg, sd, device=0, max_batch=8, precision="f16x")
        try:
            logits = eng.forward(x.cuda(), 0, len(eng.stages)).cpu()
        finally:
            eng.close()
        ref = vit_oracle.forward(x, sd, cfg)["logits"]
        per = [rel_err(logits[i:i + 1], ref[i:i + 1]) for i in range(8)]
        worst_image = max(worst_image, max(per))
        worst_batch = max(worst_batch, rel_err(logits, ref))
        print(f"f16x vit_b_16 seed {seed}: logits vs plain f32 over 8 images {rel_err(logits, ref):.2e}, worst single image {max(per):.2e}")
    print(f"f16x vit_b_16, 3 seeds x 8 images: worst batch {worst_batch:.2e}, worst image {worst_image:.2e}")
    assert worst_image <= 1e-3, (worst_batch, worst_image)


def test_split_weight_low_parts_survive(monkeypatch):
    """lo = rn16(w - hi) is stored unscaled; for |w| ~ 1e-3 it is an f16 SUBNORMAL (|lo| <= 2^-21, step 2^-24).  If the MFMA
    flushed subnormal operands the split GEMM would silently degrade to the single-pass product.  A weight-split MLP-down GEMM with
    |w| ~ 1e-3 against the f64 product of the same f16 activations: the hi-only product is 2^-12-class (3e-4 rms), hi + lo must be
    at least 5 x closer
    (measured 9 x: what is left is the f32 rounding of the residual add the product is read back through)."""
    from interactive_vit_amd.engine import Engine
    from oracle import vit_oracle
    monkeypatch.setenv("IVIT_F16X_MLP2", "1")    # the down weight as a pair whatever the model's default (ivit_split_set: bit 3)
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    gen = torch.Generator().manual_seed(7)
    for i in range(cfg.layers):
        k = f"encoder.layers.encoder_layer_{i}.mlp.3.weight"
        sd[k] = (torch.rand(sd[k].shape, generator=gen) * 2 - 1) * 1.5e-3      # |w| < 2^-9: every lo part is an f16 subnormal
    eng = Engine(cfg, sd, device=0, max_batch=2, precision="f16x")
    try:
        assert "mlp2w" in eng.split_gemms
        x = synthetic_images(2, cfg, seed=9)
        tok = vit_oracle.tokens(vit_oracle.conv_proj(vit_oracle.transform(x), sd, cfg), sd, cfg)
        u = eng.layer_tap(0, tok.cuda(), "u").cpu()          # the f16 GELU output the MLP-down GEMM multiplies, as stored
        proj = eng.layer_tap(0, tok.cuda(), "proj").cpu().double().reshape(-1, cfg.dim)   # the residual it adds to
        out = eng.layer_tap(0, tok.cuda(), "out").cpu().double().reshape(-1, cfg.dim)
        w = sd["encoder.layers.encoder_layer_0.mlp.3.weight"].double()
        b = sd["encoder.layers.encoder_layer_0.mlp.3.bias"].double()
        ud = u.double().reshape(-1, cfg.mlp)
        got = out - proj - b                                   # the product alone
        exact = ud @ w.t()
        hi_only = ud @ w.to(torch.float16).double().t()
        e_split = float((got - exact).abs().max() / exact.abs().max())
        e_hi = float((hi_only - exact).abs().max() / exact.abs().max())
        print(f"split MLP-down GEMM, |w| < 1.5e-3: hi + lo {e_split:.2e} from the f64 product, hi alone {e_hi:.2e}")
        assert e_hi > 1e-4 and e_split < e_hi / 5, (e_split, e_hi)
    finally:
        eng.close()


def test_async_host_outputs_are_lazy_and_bit_identical(small, monkeypatch):
    """on the host path every node returns at once with a lazily-synchronised tensor (PendingTensor,
    include/ivit.h: ivit_forward_host_async); the D2H copy of node k runs behind node k+1's kernels and the first torch
    operation on a tensor waits for it.  Bytes must equal the synchronous path's in every situation."""
    from interactive_vit_amd.engine import Engine, PendingTensor
    cfg, sd, eng = small
    assert eng._async
    monkeypatch.setenv("IVIT_ASYNC_OUTPUTS", "0")
    sync_eng = Engine(cfg, sd, device=0, max_batch=5)
    try:
        assert not sync_eng._async
        ns = len(eng.stages)
        img = synthetic_images(1, cfg, seed=77)[0]
        want, x = [], img
        for s in range(ns):
            x = sync_eng.forward(x, s, s + 1)
            assert not isinstance(x, PendingTensor)
            want.append(x)
        for rep in range(3):                         # rep 0: eager + graph capture; later: graph replay
            outs, x = [], img
            for s in range(ns):
                x = eng.forward(x, s, s + 1)         # returns without waiting; x is handed on by reference (chained, resident)
                assert isinstance(x, PendingTensor)
                outs.append(x)
            for s in reversed(range(ns)):            # read in any order, also the early ones last
                assert torch.equal(outs[s], want[s]), f"stage {s} (rep {rep})"
        # a pending tensor as input after ANOTHER request ran in between: the stale token falls back to an upload that is
        # stream-ordered behind the pending copy
        a = eng.forward(img, 0, 1)
        eng.forward(synthetic_images(1, cfg, seed=78)[0], 0, 2)
        assert torch.equal(eng.forward(a, 1, 2), want[1])
        # torch operations on a pending tensor give plain tensors with the finished bytes
        b = eng.forward(img, 0, 2)
        c = b * 2.0
        assert type(c) is torch.Tensor and torch.equal(c, want[1] * 2.0)
        assert b.numpy().shape == tuple(want[1].shape)
        # modified in place after it was returned: the engine must use the new bytes
        d = eng.forward(img, 0, 1)
        d.mul_(0.5)
        assert torch.equal(eng.forward(d, 1, 2), sync_eng.forward(want[0] * 0.5, 1, 2))
        # a pending tensor of ONE engine as the input of ANOTHER (two model plugins chained in a graph): nothing of the second
        # engine is ordered behind the first one's copy stream, so the binding waits before the upload
        monkeypatch.setenv("IVIT_ASYNC_OUTPUTS", "1")
        other = Engine(cfg, sd, device=0, max_batch=5)
        try:
            for _ in range(3):
                a = eng.forward(img, 0, 2)               # pending, owned by `eng`
                assert torch.equal(other.forward(a, 2, 3), want[2])
        finally:
            other.close()
    finally:
        sync_eng.close()


def test_dropped_pending_output_keeps_its_buffer_until_the_copy_has_landed(small):
    """a PendingTensor dropped before anything waited on it must not hand its page-locked block back to torch's
    pool while the engine's copy stream is still writing it.  The engine holds the buffer until its ticket has completed."""
    import gc
    from interactive_vit_amd.engine import PendingTensor
    cfg, sd, eng = small
    assert eng._async
    img = synthetic_images(1, cfg, seed=91)[0]
    want = eng.forward(img, 0, 2).clone()                      # (waits) the reference bytes
    for _ in range(4):
        out = eng.forward(img, 0, 2)
        assert isinstance(out, PendingTensor)
        ptr, nbytes, ticket = out._ivit_ptr, int(np.prod(out._ivit_shape)) * 4, out._ivit_ticket   # (no torch call: that would wait)
        del out
        gc.collect()
        assert ticket in eng._held, "the engine must hold the buffer of a ticket nobody waited for"
        # the same-sized pinned allocation that follows must NOT get the block the DMA may still be writing
        other = torch.empty(nbytes // 4, dtype=torch.float32, pin_memory=True)
        assert other.data_ptr() != ptr
        other.fill_(-7.0)
        held = eng._held[ticket]
        eng._check(eng.lib.ivit_host_wait(eng._h, __import__("ctypes").c_uint64(ticket)))
        eng._retire(ticket)
        assert ticket not in eng._held
        assert torch.equal(held.reshape(want.shape), want) and bool((other == -7.0).all())
    # the number of held buffers stays bounded when nobody ever waits
    for _ in range(eng._MAX_HELD + 8):
        eng.forward(img, 0, 1)
    assert len(eng._held) <= eng._MAX_HELD + 1


def test_a_failed_call_still_hands_the_workspaces_over():
    """a call that fails AFTER it took the shared workspaces (here: a layer tap on an fp8 engine that was
    never calibrated - run_layer refuses behind ws_acquire) must still record the hand-over event and keep its own message;
    the next call, on another stream, then runs normally."""
    from interactive_vit_amd.engine import Engine, EngineError
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    eng = Engine(cfg, sd, device=0, max_batch=2, precision="fp8")
    try:
        x = synthetic_images(2, cfg, seed=5)
        tok = torch.randn(2, cfg.tokens, cfg.dim, device="cuda")
        with pytest.raises(EngineError, match="not calibrated"):
            eng.layer_tap(0, tok, "qkv")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):                       # another stream: ordered behind the failed call's partial work by the event
            eng.calibrate_fp8(x)
            out = eng.forward(x.cuda(), 0, len(eng.stages))
        side.synchronize()
        assert torch.isfinite(out).all()
    finally:
        eng.close()


def test_checkpoint_file_through_the_engine(tmp_path):
    """A local safetensors checkpoint with mixed-precision tensors -> weights.load_state_dict_file -> ivit_set_weight: the engine's
    forward equals the one on the same values handed over as a dict, and the LayerNorm-fold guard runs on it."""
    from safetensors.torch import save_file
    from interactive_vit_amd.models.vit import HipBackend
    from interactive_vit_amd.weights import load_state_dict_file
    cfg = small_config()
    sd = init_weights(cfg, seed=21, mode="rich")
    stored = {k: v.to(torch.bfloat16 if i % 2 else torch.float32) for i, (k, v) in enumerate(sd.items())}
    path = str(tmp_path / "ckpt.safetensors")
    save_file(stored, path)
    loaded = load_state_dict_file(path, cfg)
    same = {k: v.to(torch.float32) for k, v in stored.items()}
    a = HipBackend(cfg, loaded, device=0, max_batch=2)
    b = HipBackend(cfg, same, device=0, max_batch=2)
    try:
        assert a.ln_fold_ratio is not None and a.ln_fold_ratio == b.ln_fold_ratio
        x = synthetic_images(2, cfg, seed=4)
        assert torch.equal(a.run_node("forward", x), b.run_node("forward", x))
    finally:
        a.engine.close(); b.engine.close()
    # the operator's own sample pictures and threshold decide the fold for that checkpoint: a threshold below the measured ratio drops it
    c = HipBackend(cfg, loaded, device=0, max_batch=2, calibration_images=synthetic_images(2, cfg, seed=99), ln_fold_threshold=1e-6)
    d = HipBackend(cfg, loaded, device=0, max_batch=2, calibration_images=synthetic_images(1, cfg, seed=99)[0], ln_fold_threshold=1e6)
    try:
        assert c.ln_fold_ratio > 1e-6 and not c.engine.ln_fold_for(2)
        assert d.engine.ln_fold_for(2)
    finally:
        c.engine.close(); d.engine.close()
