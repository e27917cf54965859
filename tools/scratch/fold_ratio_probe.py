import os, sys, torch
sys.path.insert(0, os.getcwd())
from interactive_vit_amd.engine import Engine
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images
cfg = VARIANTS["vit_b_16"]
sd = init_weights(cfg, seed=0, mode="spec")
for nimg, seed in ((4, 1234), (2, 7), (16, 5)):
    eng = Engine(cfg, sd, device=0, max_batch=16)
    x = synthetic_images(nimg, cfg, seed=seed, device="cuda:0")
    r = eng.calibrate_ln_fold(x)
    print(f"vit_b_16 spec weights, {nimg} images seed {seed}: centred statistic {r:.4f}, plain {eng.ln_fold_ratio_plain:.4f}, fold kept {eng.ln_fold}")
    eng.close()
