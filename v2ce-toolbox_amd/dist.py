"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl"
= RCCL over xGMI on ROCm; "gloo" in the CPU tests).

What is sharded (``pipeline.run_clip``): the 16-pair SEQUENCES of every reference batch
(v2ce.py:163-204 makes one model call per batch of ``-b`` sequences): rank r takes a contiguous share of
the batch's sequences, so BASELINE config 3 (``-b 32`` on 8 GPUs) is four sequences per GPU per batch.
All ranks walk the batches in lockstep, hence every rank applies the spectral-norm power iteration of
call k exactly when the single-process run does; the split-half kernels keep one range slot per batch
element (include/v2ce_hip.h ``absmax_batch_stride``), so a sequence's voxels are bit-identical whatever
shares its launch.  No collective on the data path.  Pano (v2ce.py:100-129) with a world that is a multiple
of the tile count: one W-tile per GPU of a tile group + ONE all-to-all per batch that re-shards W-tiles ->
frame-pairs (``tiles_to_pairs``); sequences are shared out over the groups.

The only other exchange is the variable-length gather of the packed 13-byte event records to rank 0
(north_star: "RCCL-over-xGMI gather of the final event list"), streamed batch by batch
(``StreamedGather``): RCCL has no gatherv, so it is one all_gather of byte counts plus one gather of
buffers padded to the longest rank, both on a communication stream; rank r's records of a step precede
rank r+1's in the global frame-pair order, so rank 0 appends what it receives.  Event traffic (13 B/event)
is orders of magnitude below one xGMI link: latency-, not bandwidth-bound.

``Comm`` is the seam: ``TorchComm`` (torch.distributed) is the product; ``ThreadWorld`` runs the SAME
driver code as N ranks on N threads of one process sharing one GPU (barrier-backed exchanges) -- how a
1-GPU box exercises the C3 / C4 plans with the real kernels (tests/test_gpu_multirank.py).
"""
from __future__ import annotations

import collections
import threading
from typing import Callable, List, Optional, Sequence, Tuple

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
    Returns the concatenated tensor on `dst`, None elsewhere.  (Blocking form; the drivers use
    ``StreamedGather``.)"""
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
    """One variable-length gather split in two so that no rank ever waits on its compute stream:

    ``begin``  (constructor) -- on a communication stream that waits only for the event recorded behind the
                  producer of `packed`: all_gather of the byte counts, copied to pinned host memory.
    ``finish`` -- the host waits for that copy alone (typically one step later, long done), then enqueues the
                  padded ``gather`` on the communication stream.  Returns the list of per-rank buffers on
                  `dst` (not concatenated, not waited for: they are valid on ``self.stream``), None elsewhere."""

    _streams = {}

    def __init__(self, packed: torch.Tensor, dst: int = 0, group=None):
        self.packed, self.dst, self.group = packed, dst, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = packed.is_cuda
        self.stream = None
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
            send = torch.empty(longest, dtype=torch.uint8, device=self.packed.device)
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


class StreamedGather:
    """The per-step gather of the drivers (pipeline.run_clip and bench.py share it): ``submit(packed)``
    starts the gather of this step's records (``EventGather``) and completes the PREVIOUS step's -- its byte
    counts are long on the host by then --, handing rank `dst` the per-rank buffers in rank order through
    ``on_pieces(pieces, stream)`` (`stream`: the communication stream they are valid on; None on CPU).
    ``drain()`` completes what is in flight.  Every rank must submit the same number of steps (empty
    buffers count)."""

    def __init__(self, dst: int = 0, group=None, on_pieces: Optional[Callable] = None, depth: int = 1):
        self.dst, self.group, self.on_pieces, self.depth = dst, group, on_pieces, depth
        self.inflight = collections.deque()
        self.bytes_last = 0

    def submit(self, packed: torch.Tensor) -> None:
        self.inflight.append(EventGather(packed, self.dst, self.group))
        while len(self.inflight) > self.depth:
            self._finish_one()

    def _finish_one(self) -> None:
        g = self.inflight.popleft()
        pieces = g.finish()
        if pieces is not None:
            self.bytes_last = int(sum(p.numel() for p in pieces))
            if self.on_pieces is not None:
                self.on_pieces(pieces, g.stream)

    def drain(self) -> None:
        while self.inflight:
            self._finish_one()


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


def _tile_exchange_buffers(part: torch.Tensor, widths: Sequence[int], tile_index: int):
    n = len(widths)
    P = part.shape[0]
    ranges = [shard_range(P, r, n) for r in range(n)]
    lo, hi = ranges[tile_index]
    inputs = [part[a:b].contiguous() for a, b in ranges]
    outputs = [torch.empty((hi - lo,) + tuple(part.shape[1:-1]) + (w,), dtype=part.dtype, device=part.device)
               for w in widths]
    return inputs, outputs, lo


def tiles_to_pairs(part: torch.Tensor, widths: Sequence[int], tile_index: int, group) -> Tuple[torch.Tensor, int]:
    """Re-shard one batch from W-tiles to frame-pairs (pano): this rank holds tile
    `tile_index` of all P frame-pairs, part [P,2,10,H,widths[tile_index]]; afterwards it holds ALL
    tiles, concatenated on the width, of its contiguous share of the pairs (LDATI sorts a (frame, bin)
    segment over the full width, LDATI.py:296-297).  Returns ([P_r,2,10,H,sum(widths)], first pair).
    A share may be empty (fewer pairs than tiles): the rank still takes part in the exchange."""
    inputs, outputs, lo = _tile_exchange_buffers(part, widths, tile_index)
    all_to_all_v(outputs, inputs, group)
    return torch.cat(outputs, dim=-1), lo


def fast_forward(model, global_call_index: int) -> None:
    """Advance the spectral-norm u/v of `model` so that its NEXT call is global call
    `global_call_index` of the single-process reference schedule."""
    while model.calls < global_call_index:
        model.advance_spectral_norm()


# ------------------------------------------------------------------------------------------------
# communicators
# ------------------------------------------------------------------------------------------------
class LocalComm:
    """World of one: nothing to exchange."""
    rank, world = 0, 1

    def max_float(self, v: float, device=None) -> float:
        return v

    def barrier(self) -> None:
        pass


class TorchComm:
    """torch.distributed (RCCL on the GPU box, gloo in the CPU tests)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def tile_group(self, size: int):
        return subgroup(size, self.rank, self.world)

    def tiles_to_pairs(self, part, widths, tile_index, tile_group):
        return tiles_to_pairs(part, widths, tile_index, tile_group)

    def streamed_gather(self, on_pieces, dst: int = 0) -> StreamedGather:
        return StreamedGather(dst, self.group, on_pieces)

    def max_float(self, v: float, device=None) -> float:
        t = torch.tensor([v], dtype=torch.float32, device=device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self) -> None:
        dist.barrier(group=self.group)


def default_comm(force: bool = False):
    """TorchComm when torch.distributed is initialised with more than one rank (or `force`: a world of
    one still takes the collective code path -- how a 1-GPU box runs the RCCL calls), else LocalComm."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        return TorchComm()
    return LocalComm()


class ThreadWorld:
    """N ranks as N threads of ONE process (sharing one GPU and its default stream): every exchange is a
    deposit into shared memory between two barriers.  Same driver code as under torch.distributed; used to
    run the multi-GPU plans of BASELINE configs 3 / 4 with the real kernels on a 1-GPU box and to compare
    them byte for byte with the single-rank run.  ``run(fn)`` calls fn(comm) on every rank's thread and
    returns the list of results (exceptions are re-raised)."""

    def __init__(self, world: int):
        self.world = world
        self.failed = False
        self.barriers = {}                                  # one barrier per set of members (world, tile groups)
        self.slots = {}
        self.lock = threading.Lock()

    def barrier_of(self, members: tuple) -> threading.Barrier:
        with self.lock:
            if members not in self.barriers:
                self.barriers[members] = threading.Barrier(len(members))
                if self.failed:                             # a rank has died: nobody may wait for it
                    self.barriers[members].abort()
            return self.barriers[members]

    def abort(self) -> None:
        with self.lock:
            self.failed = True
            bs = list(self.barriers.values())
        for b in bs:
            b.abort()

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)

    def run(self, fn: Callable[["ThreadComm"], object], device=None) -> list:
        results, errors = [None] * self.world, [None] * self.world
        current = torch.cuda.current_device() if (device is not None and torch.device(device).type == "cuda") else 0

        def body(r):
            try:
                if device is not None and torch.device(device).type == "cuda":
                    d = torch.device(device)
                    torch.cuda.set_device(d if d.index is not None else torch.device("cuda", current))
                results[r] = fn(self.comm(r))
            except BaseException as e:                       # noqa: BLE001 -- re-raised on the caller's thread
                errors[r] = e
                self.abort()
        threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in errors:
            if e is not None:
                raise e
        return results


class ThreadComm:
    def __init__(self, world: ThreadWorld, rank: int):
        self.w, self.rank, self.world = world, rank, world.world
        self.seq = {}

    def _exchange(self, value, members: Sequence[int]):
        """All `members` deposit `value`; returns {rank: value}.  Every member makes the same sequence of
        exchanges with that member set (the drivers do)."""
        members = tuple(members)
        n = self.seq.get(members, 0)
        self.seq[members] = n + 1
        key, bar = (members, n), self.w.barrier_of(members)
        with self.w.lock:
            self.w.slots.setdefault(key, {})[self.rank] = value
        bar.wait()
        got = dict(self.w.slots[key])
        bar.wait()
        if self.rank == members[0]:
            with self.w.lock:
                del self.w.slots[key]
        return got

    def tile_group(self, size: int):
        g = self.rank // size
        return tuple(range(g * size, (g + 1) * size))

    def tiles_to_pairs(self, part, widths, tile_index, tile_group):
        inputs, outputs, lo = _tile_exchange_buffers(part, widths, tile_index)
        got = self._exchange(inputs, tile_group)
        for j, r in enumerate(tile_group):
            outputs[j].copy_(got[r][tile_index])
        return torch.cat(outputs, dim=-1), lo

    def streamed_gather(self, on_pieces, dst: int = 0):
        comm = self

        class _G:
            bytes_last = 0

            def submit(self, packed):
                got = comm._exchange(packed, range(comm.world))
                if comm.rank == dst:
                    pieces = [got[r] for r in range(comm.world)]
                    self.bytes_last = int(sum(p.numel() for p in pieces))
                    if on_pieces is not None:
                        on_pieces(pieces, None)

            def drain(self):
                pass
        return _G()

    def max_float(self, v: float, device=None) -> float:
        return max(self._exchange(v, range(self.world)).values())

    def barrier(self) -> None:
        self.w.barrier_of(tuple(range(self.world))).wait()
