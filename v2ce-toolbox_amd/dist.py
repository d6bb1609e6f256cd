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

from typing import List, Optional, Sequence, Tuple

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


class EventGather:
    """``gather_events`` split in two so that no rank ever waits on its compute stream (bench.py's
    per-step gather; a driver that streams results to rank 0 batch by batch would use it the same way):

    ``begin``  -- on a communication stream that waits only for the event recorded behind the producer of
                  `packed`: all_gather of the byte counts, copied to pinned host memory.
    ``finish`` -- the host waits for that copy alone (typically one step later, long done), then enqueues the
                  padded ``gather`` on the communication stream.  Returns the list of per-rank buffers on
                  `dst` (not concatenated, not waited for: synchronise before reading), None elsewhere."""

    _streams = {}

    def __init__(self, packed: torch.Tensor, dst: int = 0, group=None):
        self.packed, self.dst, self.group = packed, dst, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = packed.is_cuda
        n = torch.tensor([packed.numel()], dtype=torch.int64)
        if self.cuda:
            dev = packed.device
            if dev not in EventGather._streams:
                EventGather._streams[dev] = torch.cuda.Stream(device=dev)
            self.stream = EventGather._streams[dev]
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            self.stream.wait_event(ready)
            with torch.cuda.stream(self.stream):
                packed.record_stream(self.stream)
                sizes = torch.empty(self.world, dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(sizes, n.to(dev), group=group)
                self.sizes_host = torch.empty(self.world, dtype=torch.int64).pin_memory()
                self.sizes_host.copy_(sizes, non_blocking=True)
                self.sizes_ready = torch.cuda.Event()
                self.sizes_ready.record(self.stream)
        else:
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(sizes, n, group=group)
            self.sizes_host = torch.cat(sizes)

    def finish(self):
        if self.cuda:
            self.sizes_ready.synchronize()
        sizes = [int(v) for v in self.sizes_host.tolist()]
        longest = max(max(sizes), 1)
        ctx = torch.cuda.stream(self.stream) if self.cuda else _NullCtx()
        with ctx:
            send = torch.zeros(longest, dtype=torch.uint8, device=self.packed.device)
            send[:self.packed.numel()] = self.packed
            recv = [torch.empty(longest, dtype=torch.uint8, device=self.packed.device) for _ in range(self.world)] \
                if self.rank == self.dst else None
            dist.gather(send, recv, dst=self.dst, group=self.group)
            if self.cuda:
                self.done = torch.cuda.Event()
                self.done.record(self.stream)
        self.packed = None
        if self.rank != self.dst:
            return None
        return [recv[r][:sizes[r]] for r in range(self.world)]


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_SUBGROUPS = {}


def subgroup(size: int, rank: int, world: int):
    """Process group of the `size` consecutive ranks that contain `rank` (ranks [g*size, (g+1)*size)).
    Every rank creates every group (torch.distributed requires it); cached per (size, world)."""
    key = (size, world)
    if key not in _SUBGROUPS:
        _SUBGROUPS[key] = [dist.new_group(list(range(g * size, (g + 1) * size))) for g in range(world // size)]
    return _SUBGROUPS[key][rank // size]


def all_to_all_v(outputs: List[torch.Tensor], inputs: List[torch.Tensor], group=None) -> None:
    """Variable-size all-to-all inside `group`: inputs[j] goes to group rank j, outputs[j] comes from
    it.  RCCL: one grouped send/recv (``dist.all_to_all``); gloo has no all-to-all, so the CPU tests
    take the same exchange as batched point-to-point operations."""
    if dist.get_backend(group) == "nccl":
        dist.all_to_all(outputs, inputs, group=group)
        return
    me = dist.get_rank(group)
    outputs[me].copy_(inputs[me])
    ops = []
    for j in range(dist.get_world_size(group)):
        if j == me:
            continue
        peer = dist.get_global_rank(group, j) if group is not None else j
        ops.append(dist.P2POp(dist.isend, inputs[j], peer, group=group))
        ops.append(dist.P2POp(dist.irecv, outputs[j], peer, group=group))
    for req in dist.batch_isend_irecv(ops) if ops else []:
        req.wait()


def tiles_to_pairs(part: torch.Tensor, widths: Sequence[int], tile_index: int, group) -> Tuple[torch.Tensor, int]:
    """Re-shard one batch from W-tiles to frame-pairs (pano): this rank holds tile
    `tile_index` of all P frame-pairs, part [P,2,10,H,widths[tile_index]]; afterwards it holds ALL
    tiles, concatenated on the width, of its contiguous share of the pairs (LDATI sorts a (frame, bin)
    segment over the full width, LDATI.py:296-297).  Returns ([P_r,2,10,H,sum(widths)], first pair)."""
    n = len(widths)
    P = part.shape[0]
    ranges = [shard_range(P, r, n) for r in range(n)]
    lo, hi = ranges[tile_index]
    inputs = [part[a:b].contiguous() for a, b in ranges]
    outputs = [torch.empty((hi - lo,) + tuple(part.shape[1:-1]) + (w,), dtype=part.dtype, device=part.device)
               for w in widths]
    all_to_all_v(outputs, inputs, group)
    return torch.cat(outputs, dim=-1), lo


def gather_segments(local: torch.Tensor, segments, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Gather byte buffers that consist of keyed segments: `local` = this rank's segments back to back,
    `segments` = [(order key, bytes)] in that order.  `dst` gets all segments of all ranks in key
    order (the keys are (batch, position in the batch): global frame-pair order); None elsewhere.
    One object all-gather of the (small) tables + the padded gather of ``gather_events``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tables = [None] * world
    dist.all_gather_object(tables, list(segments), group=group)
    sizes = [sum(n for _, n in t) for t in tables]
    longest = max(max(sizes), 1)
    send = torch.zeros(longest, dtype=torch.uint8, device=local.device)
    send[:local.numel()] = local
    recv = [torch.empty(longest, dtype=torch.uint8, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    pieces = []
    for r, t in enumerate(tables):
        off = 0
        for key, n in t:
            pieces.append((key, recv[r][off:off + n]))
            off += n
    pieces.sort(key=lambda kv: kv[0])
    return torch.cat([p for _, p in pieces]) if pieces else torch.empty(0, dtype=torch.uint8, device=local.device)


def fast_forward(model, global_call_index: int) -> None:
    """Advance the spectral-norm u/v of `model` so that its NEXT call is global call
    `global_call_index` of the single-process reference schedule."""
    while model.calls < global_call_index:
        model.advance_spectral_norm()
