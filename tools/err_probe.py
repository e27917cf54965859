"""Diagnostic: per-node relative error (max|d|/max|ref|) of the GPU chain vs the f32 oracle."""
import sys, torch
sys.path.insert(0, ".")
from interactive_vit_amd.vit_config import VARIANTS, test_config
from interactive_vit_amd.weights import init_weights, synthetic_images
from interactive_vit_amd.engine import Engine
from oracle import vit_oracle as vo

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return float((a-b).abs().max()/b.abs().max()), float((a-b).norm()/b.norm())

for name, mode in (("vit_test","rich"), ("vit_ti_16","rich"), ("vit_b_16","spec"), ("vit_b_16","rich")):
    cfg = test_config() if name == "vit_test" else VARIANTS[name]
    sd = init_weights(cfg, 0, mode)
    eng = Engine(cfg, sd, 0, 2)
    x = synthetic_images(2, cfg, 1234)
    acts = vo.forward(x, sd, cfg, keep=True)
    cur_chain = x.cuda(); prev_ref = x
    print(f"== {name} {mode}")
    for s in vo.node_suffixes(cfg):
        alone = eng.run_node(s, prev_ref.cuda())
        cur_chain = eng.run_node(s, cur_chain)
        ra = rel(alone, acts[s]); rc = rel(cur_chain, acts[s])
        print(f"{s:22s} alone max {ra[0]:.2e} l2 {ra[1]:.2e} | chained max {rc[0]:.2e} l2 {rc[1]:.2e} | ref max {float(acts[s].abs().max()):.3f}")
        prev_ref = acts[s]
    eng.close()
