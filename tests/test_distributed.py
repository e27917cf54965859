"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, run the (oracle-backed) forward
on their shard and reassemble with the ONE all-gather of interactive_vit_amd.sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from interactive_vit_amd.sharding import (all_gather_outputs, pack_outputs, shard_range, shard_sizes,
                                           split_outputs)
from interactive_vit_amd.vit_config import test_config as small_config
from interactive_vit_amd.weights import init_weights, synthetic_images


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 2, 7, 64, 2048):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)       # BASELINE config 4: 256 images per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vit_oracle as vo
        torch.set_num_threads(1)
        cfg = small_config()
        sd = init_weights(cfg, seed=3, mode="rich")          # weights replicated on every rank
        x = synthetic_images(total, cfg, seed=21)            # the global batch, same on every rank
        b0, b1 = shard_range(total, rank, world)
        acts = vo.forward(x[b0:b1], sd, cfg)
        local = pack_outputs(acts["logits"], acts["cls"])
        gathered = all_gather_outputs(local, total)
        logits, cls = split_outputs(gathered, cfg.classes)
        q.put((rank, logits.clone(), cls.clone()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])     # equal shards; ragged shards (3 + 2)
def test_two_rank_gloo_all_gather_reassembles_the_batch(total):
    from oracle import vit_oracle as vo
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = small_config()
    sd = init_weights(cfg, seed=3, mode="rich")
    full = vo.forward(synthetic_images(total, cfg, seed=21), sd, cfg)
    for rank, logits, cls in results:
        assert logits.shape == (total, cfg.classes) and cls.shape == (total, cfg.dim)
        assert torch.allclose(logits, full["logits"], atol=1e-5)     # every rank holds the whole batch
        assert torch.allclose(cls, full["cls"], atol=1e-5)
