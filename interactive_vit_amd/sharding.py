"""Batch sharding over the GPUs of one node and the single all-gather that reassembles outputs.

A ViT forward has no cross-image operation, so images shard by batch with replicated weights
(SURVEY 8(e)): one process per GPU, rank r owns the contiguous slice ``shard_range(B, r, W)``.
The only collective of the path is ONE all-gather per batch of each rank's ``[b_local, classes + D]``
block (logits and class-token features packed side by side so that it really is a single
collective) - RCCL over xGMI under ``backend="nccl"``, gloo in the CPU tests.  The payload is
small (256 x 1768 f32 = 1.8 MB per rank at B=2048), latency-bound; it is issued on the compute
stream right after the head GEMM.

The reference has no counterpart (no torch.distributed anywhere).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of rank's contiguous shard; the first ``total % world`` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def pack_outputs(logits: torch.Tensor, cls: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[b, classes] and [b, D] -> one [b, classes + D] block (the all-gather payload)."""
    if out is None:
        return torch.cat([logits, cls], dim=1)
    c = logits.shape[1]
    out[:, :c].copy_(logits)
    out[:, c:].copy_(cls)
    return out


def all_gather_outputs(local: torch.Tensor, total: int, group=None,
                       out: Optional[torch.Tensor] = None, async_op: bool = False):
    """ONE collective: every rank ends with the [total, width] outputs of the whole batch, in image
    order.  Equal shards use ``all_gather_into_tensor`` directly; ragged shards are padded to the
    largest shard for the collective and compacted afterwards (still one collective).

    ``async_op=True`` (equal shards, ``out`` given): returns the work handle instead of the tensor, so
    that the collective overlaps the next batch's compute; ``out`` is valid after ``handle.wait()``."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(total, world)
    assert local.shape[0] == sizes[rank], f"rank {rank} holds {local.shape[0]} rows, expected {sizes[rank]}"
    width = local.shape[1]
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty((total, width), dtype=local.dtype, device=local.device)
        if async_op:
            return dist.all_gather_into_tensor(out, local.contiguous(), group=group, async_op=True)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    if async_op:
        raise ValueError("async_op needs equal shards")
    big = max(sizes)
    padded = torch.zeros((big, width), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]].copy_(local)
    gathered = torch.empty((world * big, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    parts = [gathered[r * big:r * big + sizes[r]] for r in range(world)]
    res = torch.cat(parts, dim=0)
    if out is not None:
        out.copy_(res)
        return out
    return res


def split_outputs(gathered: torch.Tensor, classes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    return gathered[:, :classes], gathered[:, classes:]
