"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards over independent 16-pair SEQUENCES (v2ce.py:163-204 processes them in a plain loop)
with no collective on the data path; the only exchange is the final variable-length gather of the
packed 13-byte event records to rank 0 (north_star: "RCCL-over-xGMI gather of the final event
list").  RCCL has no gatherv, so it is one all_gather of byte counts plus one gather of buffers
padded to the longest rank.  Event traffic (13 B/event) is orders of magnitude below one xGMI
link, so this is latency- not bandwidth-bound.

Spectral-norm state under sharding: the reference applies one power iteration per
model call, so a replica that emulates global call index k must have applied k iterations before
its call; ``fast_forward`` does that (the u/v trajectory never depends on the input data).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_items` for `rank` (earlier ranks take the remainder), so the
    rank-order concatenation of per-rank outputs is the global order."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_events(packed: torch.Tensor, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather variable-length uint8 record buffers to `dst`, concatenated in rank order.
    Returns the concatenated tensor on `dst`, None elsewhere.

    RCCL has no gatherv: one all_gather of the byte counts, then ONE `gather` of buffers padded to
    the longest rank (event traffic is 13 B/event -- far below an xGMI link -- so the padding is
    cheaper than a second protocol)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return packed
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([packed.numel()], dtype=torch.int64, device=packed.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=packed.device) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    longest = max(max(sizes), 1)
    send = torch.zeros(longest, dtype=torch.uint8, device=packed.device)
    send[:packed.numel()] = packed
    recv = [torch.empty(longest, dtype=torch.uint8, device=packed.device) for _ in range(world)] \
        if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][:sizes[r]] for r in range(world)])


def fast_forward(model, global_call_index: int) -> None:
    """Advance the spectral-norm u/v of `model` so that its NEXT call is global call
    `global_call_index` of the single-process reference schedule."""
    while model.calls < global_call_index:
        model.advance_spectral_norm()
