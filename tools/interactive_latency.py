"""Latency of the interactive path: one /compute request (wire bytes in -> wire bytes out) for a
single 224x224 image through the whole ViT-B/16 node chain on the GPU plugin, and through the
fused `forward` node alone.  Host buffers cross PCIe here (this is NOT bench.py's `value`)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from interactive_vit_amd import context as ctxmod
from interactive_vit_amd.context import Context, Model
from interactive_vit_amd.graph import Pinout
from interactive_vit_amd.message import encode_request, decode_response
from interactive_vit_amd.models.vit import HipBackend, make_vit_model_class
from interactive_vit_amd.views import compute_bytes
from interactive_vit_amd.vit_config import VARIANTS
from interactive_vit_amd.weights import init_weights, synthetic_images

base = tempfile.mkdtemp(); os.makedirs(os.path.join(base, "static", "graphs")); ctxmod.set_base_dir(base)
cfg = VARIANTS[sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"]
sd = init_weights(cfg, 0)
vit = make_vit_model_class(Model, Pinout)(cfg, HipBackend(cfg, sd, device=0, max_batch=1))
ctx = Context(); vit.register(ctx)
img = synthetic_images(1, cfg, 1)[0]
chain = vit.chain_node_names()
def req_chain():
    nodes = [{"endpoint": n, "params": {}} for n in chain]
    edges = [{"tensor": 0, "out_port": {"node": 0, "channel": "o"}}] + [
        {"in_port": {"node": i, "channel": "o"}, "out_port": {"node": i + 1, "channel": "o"}} for i in range(len(chain) - 1)]
    return encode_request(nodes, edges, [img])
def req_fwd():
    return encode_request([{"endpoint": f"{cfg.name}:forward", "params": {}}], [{"tensor": 0, "out_port": {"node": 0, "channel": "o"}}], [img])
for label, mk in (("node chain (%d nodes, every output returned)" % len(chain), req_chain), ("fused forward node", req_fwd)):
    body = mk()
    # warm-up until the host-side pools are populated (page-locked output buffers of 18 nodes x the requests whose responses are
    # still alive, the allocator's thresholds): the first ~20 requests take 2-9 ms, each new page-locked block is a driver call
    for _ in range(40): st, out = compute_bytes(body, ctx); assert st == 200
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); st, out = compute_bytes(body, ctx); ts.append(time.perf_counter() - t0)
    order_ts = list(ts)
    ts.sort()
    print(f"{cfg.name} {label}: median {ts[len(ts)//2]*1e3:.2f} ms, p10 {ts[len(ts)//10]*1e3:.2f} ms, p90 {ts[len(ts)*9//10]*1e3:.2f} ms, request {len(body)/1e6:.2f} MB, response {len(out)/1e6:.2f} MB")
    print("   in order (ms):", " ".join(f"{t*1e3:.2f}" for t in order_ts))

# ---- where a chain request spends its time (decode / each node / encode), median of 30
from interactive_vit_amd.message import Request, Response
body = req_chain()
acc = {}
for it in range(35):
    t0 = time.perf_counter(); rq = Request(); rq.decode(body); g = rq.graph; t1 = time.perf_counter()
    per = []
    for n in g.order():
        a = time.perf_counter(); n.set_pinout(ctx.get_node(n.name).compute(n.params, n.get_pinin())); per.append((n.name, time.perf_counter() - a))
    t2 = time.perf_counter(); out = Response(g).encode(); t3 = time.perf_counter()
    if it >= 5:
        acc.setdefault("decode", []).append(t1 - t0); acc.setdefault("encode", []).append(t3 - t2)
        for name, dt in per: acc.setdefault(name.split(":")[1], []).append(dt)
med = lambda v: sorted(v)[len(v) // 2] * 1e3
print("breakdown (ms):", ", ".join(f"{k} {med(v):.3f}" for k, v in acc.items()))

# ---- the same request, asynchronous as compute_bytes runs it: enqueue of all nodes, then where Response.encode waits
import interactive_vit_amd.message as msg
acc2 = {"decode": [], "enqueue all nodes": [], "encode: waiting for tensors": [], "encode: join + headers": []}
orig = msg._as_wire_f32
for it in range(35):
    waits = [0.0]
    def timed(t, _w=waits):
        a = time.perf_counter(); r = orig(t); _w[0] += time.perf_counter() - a; return r
    msg._as_wire_f32 = timed
    t0 = time.perf_counter(); rq = Request(); rq.decode(body); t1 = time.perf_counter()
    ctx.compute(rq.graph); t2 = time.perf_counter()
    out = Response(rq.graph).encode(); t3 = time.perf_counter()
    msg._as_wire_f32 = orig
    if it >= 5:
        acc2["decode"].append(t1 - t0); acc2["enqueue all nodes"].append(t2 - t1)
        acc2["encode: waiting for tensors"].append(waits[0]); acc2["encode: join + headers"].append(t3 - t2 - waits[0])
print("asynchronous request (ms):", ", ".join(f"{k} {med(v):.3f}" for k, v in acc2.items()))
