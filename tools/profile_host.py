"""cProfile of the host side of a node-chain request (where the ~40 us per node go).  Measurement aid only."""
import cProfile, pstats, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactive_vit_amd import context as ctxmod
from interactive_vit_amd.context import Context, Model
from interactive_vit_amd.graph import Pinout
from interactive_vit_amd.message import encode_request
from interactive_vit_amd.models.vit import HipBackend, make_vit_model_class
from interactive_vit_amd.views import compute_bytes
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images
base = tempfile.mkdtemp(); os.makedirs(os.path.join(base, "static", "graphs")); ctxmod.set_base_dir(base)
cfg = VARIANTS["vit_b_16"]
vit = make_vit_model_class(Model, Pinout)(cfg, HipBackend(cfg, init_weights(cfg, 0), device=0, max_batch=1))
ctx = Context(); vit.register(ctx)
img = synthetic_images(1, cfg, 1)[0]
chain = vit.chain_node_names()
nodes = [{"endpoint": n, "params": {}} for n in chain]
edges = [{"tensor": 0, "out_port": {"node": 0, "channel": "o"}}] + [
    {"in_port": {"node": i, "channel": "o"}, "out_port": {"node": i + 1, "channel": "o"}} for i in range(len(chain) - 1)]
body = encode_request(nodes, edges, [img])
for _ in range(50): compute_bytes(body, ctx)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): compute_bytes(body, ctx)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
