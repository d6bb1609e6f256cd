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
import os
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


class RankFailure(RuntimeError):
    """Raised on EVERY rank of a streamed exchange when some rank reported a failure for that step (its byte count
    travels as -1): the ranks stop at the same step instead of waiting for a peer that has left."""

    def __init__(self, ranks):
        super().__init__(f"rank(s) {list(ranks)} failed during the clip; every rank stops at this step")
        self.ranks = list(ranks)


_COMM_STREAMS = {}


def comm_stream(dev):
    if dev not in _COMM_STREAMS:
        _COMM_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return _COMM_STREAMS[dev]


_PIN_RING = {"i": 0, "bufs": {}}


def _pinned_pair(world: int):
    """(source [1], result [world]) int64 in pinned memory from a ring of 16 per world size: at most a few exchanges are
    in flight at once (the drivers complete exchange k before they start k + 2)."""
    ring = _PIN_RING["bufs"].setdefault(world, [])
    if len(ring) < 16:
        ring.append((torch.empty(1, dtype=torch.int64).pin_memory(), torch.empty(world, dtype=torch.int64).pin_memory()))
        return ring[-1]
    _PIN_RING["i"] = (_PIN_RING["i"] + 1) % 16
    return ring[_PIN_RING["i"]]


class SizeExchange:
    """all_gather of ONE int64 per rank (a step's byte count, or -1 = "this rank has failed").  On a HIP device it
    runs on the communication stream behind an event recorded on the producer's stream; the value travels from PINNED
    host memory (a pageable source would make the copy -- and with it the host -- wait for the stream) and the
    result comes back into pinned memory: ``result()`` waits for that copy alone, typically one step later."""

    def __init__(self, value: int, device, group=None, after_stream=None, wait: bool = True):
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            self.stream = comm_stream(self.device)
            if wait:                                          # (the value is host-known: callers that move no device data behind
                ready = torch.cuda.Event()                    # this exchange pass wait=False, so that a failure flag still travels
                ready.record(after_stream if after_stream is not None else torch.cuda.current_stream(self.device))
                self.stream.wait_event(ready)                 # when the compute stream is stuck)
            self.src, self.host = _pinned_pair(self.world)   # recycled by hand: a fresh pin_memory() is a hipHostMalloc,
            self.src[0] = int(value)                          # which synchronises the device in the middle of the pipeline
            with torch.cuda.stream(self.stream):
                n = self.src.to(self.device, non_blocking=True)
                sizes = torch.empty(self.world, dtype=torch.int64, device=self.device)
                dist.all_gather_into_tensor(sizes, n, group=group)
                self.host.copy_(sizes, non_blocking=True)
                self.ready = torch.cuda.Event()
                self.ready.record(self.stream)
        else:
            self.stream = None
            sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
            dist.all_gather(sizes, torch.tensor([int(value)], dtype=torch.int64), group=group)
            self.host = torch.cat(sizes)

    def result(self) -> List[int]:
        if self.cuda:
            self.ready.synchronize()
        return [int(v) for v in self.host.tolist()]


def check_sizes(sizes: Sequence[int]) -> None:
    bad = [r for r, v in enumerate(sizes) if v < 0]
    if bad:
        raise RankFailure(bad)


class EventGather:
    """One variable-length gather split in two so that no rank ever waits on its compute stream:

    ``begin``  (constructor) -- ``SizeExchange`` of the byte counts on the communication stream.
    ``finish`` -- the host waits for the counts alone (typically one step later, long done), then enqueues the
                  padded ``gather`` on the communication stream.  Returns the list of per-rank buffers on
                  `dst` (not concatenated, not waited for: they are valid on ``self.stream``), None elsewhere.
    `pool`: a dict the caller keeps between steps: the send buffer and rank `dst`'s receive buffers live there and
    are re-allocated only when a step outgrows them (25 % head-room), so a pano-sized step (1.2 GB per rank) does not
    take world x 1.2 GB of fresh allocator blocks per step in flight."""

    def __init__(self, packed: torch.Tensor, dst: int = 0, group=None, failed: bool = False, pool: Optional[dict] = None):
        self.packed, self.dst, self.group = packed, dst, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cuda = packed.is_cuda
        self.pool = pool if pool is not None else {}
        # the byte counts do not wait for the compute stream (they are host-known, and a failure flag must always get through);
        # the copy of `packed` in finish() does
        self.sizes = SizeExchange(-1 if failed else packed.numel(), packed.device, group, wait=False)
        self.stream = self.sizes.stream
        self.ready = None
        if self.cuda:
            self.ready = torch.cuda.Event()
            self.ready.record(torch.cuda.current_stream(packed.device))
            packed.record_stream(self.stream)

    def finish(self):
        sizes = self.sizes.result()
        check_sizes(sizes)
        longest = max(max(sizes), 1)
        ctx = torch.cuda.stream(self.stream) if self.cuda else _NullCtx()
        with ctx:
            if self.ready is not None:
                self.stream.wait_event(self.ready)             # `packed` is complete on its producer's stream
            reads_done = self.pool.pop("reads_done", None)     # the consumer of the previous step's pieces (reused buffers)
            if reads_done is not None and self.cuda:
                self.stream.wait_event(reads_done)
            cap = self.pool.get("cap", 0)
            if cap < longest:
                cap = int(longest * 1.25) + 4096
                for k in ("send", "recv"):
                    self.pool.pop(k, None)
                self.pool["cap"] = cap
                self.pool["send"] = torch.empty(cap, dtype=torch.uint8, device=self.packed.device)
                if self.rank == self.dst:
                    self.pool["recv"] = torch.empty(self.world * cap, dtype=torch.uint8, device=self.packed.device)
            send = self.pool["send"][:longest]
            send[:self.packed.numel()] = self.packed
            recv = None
            if self.rank == self.dst:
                recv = [self.pool["recv"][r * longest:(r + 1) * longest] for r in range(self.world)]
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
    """The per-step DEVICE gather (``gather='device'``: the records of every rank on rank `dst`'s GPU): ``submit(packed)``
    starts the gather of this step's records (``EventGather``) and completes the PREVIOUS step's -- its byte
    counts are long on the host by then --, handing rank `dst` the per-rank buffers in rank order through
    ``on_pieces(pieces, stream)`` (`stream`: the communication stream they are valid on; None on CPU).  The receive
    buffers are reused from step to step: `on_pieces` returns a HIP event recorded behind its (asynchronous) reads of the
    pieces, and the next gather waits for it on the communication stream.
    ``drain()`` completes what is in flight.  Every rank must submit the same number of steps (empty
    buffers count; ``failed=True`` makes every rank raise ``RankFailure`` at that step)."""

    def __init__(self, dst: int = 0, group=None, on_pieces: Optional[Callable] = None, depth: int = 1):
        self.dst, self.group, self.on_pieces, self.depth = dst, group, on_pieces, depth
        self.inflight = collections.deque()
        self.bytes_last = 0
        self.pool = {}
        self.error = None                                      # an error of THIS rank's consumer (rank dst's sink: full disk, ...)
        self.device = None

    def submit(self, packed: torch.Tensor, failed: bool = False) -> None:
        # a local consumer error is reported like a stage-2 failure: with the next byte count, so that every rank stops at
        # the same step instead of waiting for a rank that has left
        self.device = packed.device
        self.inflight.append(EventGather(packed, self.dst, self.group, failed=failed or self.error is not None, pool=self.pool))
        while len(self.inflight) > self.depth:
            self._finish_one()

    def _finish_one(self) -> None:
        g = self.inflight.popleft()
        try:
            pieces = g.finish()
        except RankFailure as rf:
            if self.error is not None:                         # this rank is (one of) the failed ones: its own error
                raise self.error from rf
            raise
        if pieces is not None:
            self.bytes_last = int(sum(p.numel() for p in pieces))
            if self.on_pieces is not None and self.error is None:
                try:
                    ev = self.on_pieces(pieces, g.stream)      # a HIP event behind the consumer's reads of the pieces, or None
                except RankFailure:
                    raise
                except Exception as e:                         # noqa: BLE001 -- re-raised at the next collective point, on every rank
                    self.error, ev = e, None
                if ev is not None:
                    self.pool["reads_done"] = ev

    def drain(self) -> None:
        while self.inflight:
            self._finish_one()
        if self.device is not None:
            # the last step's consumer may have failed too: one more flag exchange, so that no rank walks on into the clip's
            # closing collectives while rank dst raises
            flags = SizeExchange(-1 if self.error is not None else 0, self.device, self.group, wait=False).result()
            try:
                check_sizes(flags)
            except RankFailure as rf:
                if self.error is not None:
                    raise self.error from rf
                raise

    def finalize(self):
        return None


_SLOT_CACHE, _SLOT_LOCK = [], threading.Lock()


def default_gather_mode(world: int) -> str:
    """How the ranks' records reach host memory unless the caller says otherwise: 'device' -- RCCL gather of the final event
    list to rank 0's HBM (north_star's exchange), rank 0 downloads everything over its one PCIe link (~57 GB/s measured) -- at
    every world size.  'host' (V2CE_GATHER=host, or gather='host') -- every rank writes its own records into the registered
    shared segment over its own link -- is the remedy reasoned for eight ranks, where the e2e regime's 8 x 7.3 GB/s of records
    no longer fit rank 0's link; it has run at world 1 (RCCL) and on gloo only, never with eight processes
    page-locking one tmpfs segment, so it stays opt-in until a run on an 8-GPU node has shown both modes (ADVICE r5;
    bench.py --gpus 8 prints both)."""
    env = os.environ.get("V2CE_GATHER")
    if env:
        return env
    return "device"


class RegisteredSegment:
    """The first `nbytes` of a host file (tmpfs) mapped and page-locked for the GPU (hipHostRegister): ``tensor`` is a uint8 view
    a copy-out stream can write with ``copy_(..., non_blocking=True)``.  Page-locking costs ~70 us per MB: a process that runs
    several clips into one segment (bench.py) keeps the object and hands it to every HostDirectGather."""

    def __init__(self, path: str, nbytes: int, page_lock: bool = True):
        """page_lock=False (CPU ranks): the plain shared mapping -- the same window arithmetic, a memcpy instead of a DMA."""
        import mmap
        self.path, self.nbytes, self.locked = path, int(nbytes), False
        fd = os.open(path, os.O_RDWR)
        try:
            self.map = mmap.mmap(fd, self.nbytes)
        finally:
            os.close(fd)
        self.tensor = torch.frombuffer(self.map, dtype=torch.uint8)
        if page_lock:
            rc = torch.cuda.cudart().cudaHostRegister(self.tensor.data_ptr(), self.nbytes, 0)
            if int(rc) != 0:
                self.tensor = None
                self.map.close()
                raise RuntimeError(f"hipHostRegister failed with code {int(rc)}")
            self.locked = True

    def close(self) -> None:
        if self.tensor is not None:
            if self.locked:
                try:
                    torch.cuda.cudart().cudaHostUnregister(self.tensor.data_ptr())
                except Exception:                               # noqa: BLE001
                    pass
            self.tensor = None
            try:
                self.map.close()
            except (BufferError, ValueError):                   # (a view still alive: the mapping goes with it)
                pass


class HostDirectGather:
    """The per-step exchange of the drivers when the records are wanted in HOST memory (``gather='host'``; the default of
    pipeline.run_clip and bench.py is 'device'): no record crosses a GPU-GPU link or rank 0's PCIe link.  Every rank must be
    able to open `path` (one host: pipeline.run_clip checks that before it chooses this mode).  Per step the ranks
    exchange only their byte counts (``SizeExchange``: one all_gather of one int64 over RCCL); every rank then knows the
    byte offset of its piece in the clip's record stream -- records of step k precede step k+1's, rank r's precede rank
    r+1's inside a step: frame-pair order -- downloads its piece over ITS OWN PCIe link into a pinned staging buffer and
    writes it at that offset of one shared file (``os.pwrite``; the streamed ``.npz`` itself, or a /dev/shm segment
    that rank 0 maps and returns as the array).  At N = 8 the device gather would push 8 x 7.3 GB/s through rank 0's
    one PCIe link (~57 GB/s measured): exactly the link's capacity; here every link carries its own
    7.3 GB/s.  ``finalize()`` (collective) returns (total bytes, CRC-32 of the stream or None) on rank 0."""

    def __init__(self, comm, device, path: str, data_start: int, need_crc: bool, slots: int = 3, depth: int = 1,
                 registered_bytes: int = 0, segment: Optional[RegisteredSegment] = None):
        """registered_bytes > 0 (GPU ranks, no CRC wanted): the first `registered_bytes` of the segment -- which rank 0 has
        sized accordingly -- are mapped and page-locked by EVERY rank (hipHostRegister of the shared mapping), and a piece that
        ends inside them goes from the GPU straight to its place in host memory on the copy-out stream: no staging buffer, no
        CPU copy, no writer of the inode to serialise behind (what the 4-6 GB/s of the pwrite path were).  Pieces
        beyond the registered window take the staging + pwrite path (the file grows).  Page-locking costs ~70 us per MB per
        rank, once per segment; `segment`: an already registered mapping of `path` (kept by the caller between clips)."""
        import queue
        import threading
        self.comm, self.device = comm, torch.device(device)
        self.cuda = self.device.type == "cuda"
        # construction is collective: a rank that cannot open the shared file must not leave its peers in the first exchange
        self.fd, open_err = None, None
        try:
            self.fd = os.open(path, os.O_RDWR)
        except OSError as e:
            open_err = e
        errs = comm.all_gather_object(None if open_err is None else repr(open_err))
        bad = [r for r, e in enumerate(errs) if e is not None]
        if bad:
            if self.fd is not None:
                os.close(self.fd)
                self.fd = None
            self.thread = None
            if open_err is not None:
                raise open_err
            raise RankFailure(bad)
        self.data_start, self.need_crc, self.depth = int(data_start), need_crc, depth
        self.reg_bytes, self.reg_seg, self.reg_own, self.reg_tensor = 0, None, False, None
        if segment is not None and self.cuda and not need_crc and data_start == 0:
            self.reg_seg, self.reg_tensor, self.reg_bytes = segment, segment.tensor, segment.nbytes
        elif registered_bytes > 0 and not need_crc and data_start == 0:
            try:                                               # (CPU ranks: the shared mapping without the page lock)
                self.reg_seg = RegisteredSegment(path, int(registered_bytes), page_lock=self.cuda)
                self.reg_own, self.reg_tensor, self.reg_bytes = True, self.reg_seg.tensor, int(registered_bytes)
            except Exception as e:                              # noqa: BLE001 -- the staging path serves everything
                import logging
                logging.getLogger("V2CE").warning("gather='host': the shared segment could not be registered (%s): staging + pwrite", e)
                self.reg_seg, self.reg_tensor, self.reg_bytes = None, None, 0
        self.dma_bytes = 0                                     # bytes that went GPU -> segment directly
        self.base = 0                                          # bytes of all ranks' completed steps
        self.ring_bytes = 0                                    # > 0 (bench.py only): offsets wrap, the segment stays bounded
        self.bytes_last = 0
        self.inflight = collections.deque()
        self.pieces = []                                       # this rank's (crc, nbytes) per step
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.free = queue.Queue()
        for _ in range(slots):                                 # pinned staging buffers are kept between clips (page-locking
            with _SLOT_LOCK:                                   # 150 MiB costs ~80 ms: tools/shm_write_probe.py)
                self.free.put(_SLOT_CACHE.pop() if _SLOT_CACHE else [None])
        self.work = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._writer, daemon=True)
        self.thread.start()

    def _write_chunk(self, mv, file_off):
        import zlib
        pos, n = 0, len(mv)
        while pos < n:                                         # (pwrite may write less than asked)
            pos += os.pwrite(self.fd, mv[pos:], file_off + pos)
        return zlib.crc32(mv) if self.need_crc else 0

    def _writer(self):
        while True:
            item = self.work.get()
            if item is None:
                return
            slot, n, done, off, _keep, host = item
            try:
                if self.error is None and n < 0:
                    if done is not None:
                        done.synchronize()                     # a piece that went by DMA: its bytes are in the segment now
                    self.pieces.append((0, -n))
                elif self.error is None and n:
                    if done is not None:
                        done.synchronize()
                    mv = memoryview(slot[0][:n].numpy() if host is None else host).cast("B")
                    # ONE writer per rank: the kernel serialises the writers of an inode, so more threads (or more ranks) do
                    # not add up -- 6.2 GB/s with one thread, 3.9 GB/s with four (tools/shm_write_probe.py)
                    crc = self._write_chunk(mv, self.data_start + off)
                    self.pieces.append((crc, n))
                else:
                    self.pieces.append((0, 0))
            except BaseException as e:                         # noqa: BLE001 -- re-raised on the caller's thread
                self.error = e
            finally:
                if slot is not None:
                    self.free.put(slot)

    def submit(self, packed: torch.Tensor, failed: bool = False, keep=()) -> None:
        # an error of this rank's writer thread (ENOSPC on /dev/shm, a full disk under the .npz.part) is reported like a
        # stage-2 failure: -1 with the next byte count, every rank stops at that step (raising it here, on this
        # rank alone, left the peers in the next collective until the watchdog fired)
        n = int(packed.numel())
        # (wait=False: the count is host-known and the copies are ordered by their own `done` / `ready` events, so a failure flag
        # still travels when the compute stream is stuck -- like StreamedGather and EventGather)
        ex = self.comm.all_gather_int(-1 if (failed or self.error is not None) else n, packed.device, wait=False)
        slot, done, host = None, None, None
        if n and packed.is_cuda and self.reg_bytes:
            # registered window: the copy itself waits for the byte counts (the piece's offset) in _finish_one; only the event
            # behind this step's kernels is taken here
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(packed.device))
            host = "dma"
        elif n and packed.is_cuda:
            slot = self.free.get()                             # blocks while every staging buffer waits for the file
            if slot[0] is None or slot[0].numel() < n:
                slot[0] = torch.empty(int(n * 1.25) + (1 << 20), dtype=torch.uint8, pin_memory=True)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(packed.device))
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ready)
                slot[0][:n].copy_(packed, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.copy_stream)
        elif n and self.reg_bytes:
            host = "dma"                                       # CPU rank with a window: placed (or written) once the offset is known
        elif n:
            host = packed.numpy()
        self.inflight.append((ex, slot, n, done, (packed, keep), host))
        while len(self.inflight) > self.depth:
            self._finish_one()

    def _finish_one(self) -> None:
        ex, slot, n, done, keep, host = self.inflight.popleft()
        try:
            sizes = ex.result()
            check_sizes(sizes)
        except BaseException as e:
            if slot is not None:
                self.free.put(slot)
            if isinstance(e, RankFailure) and self.error is not None:
                raise self.error from e                        # this rank's own writer error
            raise
        off = self.base + sum(sizes[:self.comm.rank])
        self.base += sum(sizes)
        self.bytes_last = int(sum(sizes))
        off = off % self.ring_bytes if self.ring_bytes else off
        if isinstance(host, str):                              # "dma": straight into the registered window, or staged after all
            packed = keep[0]
            if not packed.is_cuda:                             # CPU rank: a copy into the shared mapping, or the pwrite path beyond it
                if off + n <= self.reg_bytes:
                    a = self.data_start + off
                    self.reg_tensor[a:a + n].copy_(packed)
                    self.dma_bytes += n
                    self.work.put((None, -n, None, off, keep, None))
                else:
                    self.work.put((None, n, None, off, keep, packed.numpy()))
                return
            if off + n <= self.reg_bytes:
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(done)
                    a = self.data_start + off
                    self.reg_tensor[a:a + n].copy_(packed, non_blocking=True)
                    packed.record_stream(self.copy_stream)
                    fin = torch.cuda.Event()
                    fin.record(self.copy_stream)
                self.dma_bytes += n
                self.work.put((None, -n, fin, off, keep, None))    # (n < 0: a DMA piece -- the writer only waits for it and keeps `keep` alive)
                return
            slot = self.free.get()
            if slot[0] is None or slot[0].numel() < n:
                slot[0] = torch.empty(int(n * 1.25) + (1 << 20), dtype=torch.uint8, pin_memory=True)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(done)
                slot[0][:n].copy_(packed, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.copy_stream)
            host = None
        self.work.put((slot, n, done, off, keep, host))

    def drain(self) -> None:
        while self.inflight:
            self._finish_one()

    def close(self) -> None:
        """Stop the writer thread and close the file (also on the error path)."""
        if self.thread is not None:
            self.work.put(None)
            self.thread.join()
            self.thread = None
        if self.reg_tensor is not None:
            if self.copy_stream is not None:
                self.copy_stream.synchronize()
            self.reg_tensor = None
            if self.reg_own:
                self.reg_seg.close()
            self.reg_seg = None
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None
            with _SLOT_LOCK:                                   # the staging buffers go back to the process-wide cache
                while not self.free.empty() and len(_SLOT_CACHE) < 8:
                    _SLOT_CACHE.append(self.free.get_nowait())

    def finalize(self):
        self.drain()
        self.close()                                           # (joins the writer: every write of this rank is done or has failed)
        # also the barrier behind every rank's last write; a late writer error travels with it, so that all ranks raise
        got = self.comm.all_gather_object((self.pieces, None if self.error is None else repr(self.error)))
        bad = [r for r, (_, e) in enumerate(got) if e is not None]
        if self.error is not None:
            raise self.error
        if bad:
            raise RankFailure(bad)
        lists = [p for p, _ in got]
        if self.comm.rank != 0:
            return None
        total, crc = 0, 0
        if self.need_crc:
            from .npz_stream import crc32_combine
        for k in range(len(lists[0])):
            for r in range(self.comm.world):
                c, n = lists[r][k]
                if self.need_crc and n:
                    crc = crc32_combine(crc, c, n)
                total += n
        assert total == self.base, (total, self.base)
        return total, (crc if self.need_crc else None)


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

    def all_gather_object(self, obj) -> list:
        return [obj]

    def broadcast_object(self, obj, src: int = 0):
        return obj

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

    def all_gather_int(self, value: int, device, wait: bool = True) -> SizeExchange:
        return SizeExchange(value, device, self.group, wait=wait)

    def all_gather_object(self, obj) -> list:
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def broadcast_object(self, obj, src: int = 0):
        box = [obj]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

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

            def submit(self, packed, failed=False):
                got = comm._exchange(None if failed else packed, range(comm.world))
                bad = [r for r in range(comm.world) if got[r] is None]
                if bad:
                    raise RankFailure(bad)
                if comm.rank == dst:
                    pieces = [got[r] for r in range(comm.world)]
                    self.bytes_last = int(sum(p.numel() for p in pieces))
                    if on_pieces is not None:
                        on_pieces(pieces, None)

            def drain(self):
                pass

            def finalize(self):
                return None
        return _G()

    def all_gather_int(self, value: int, device, wait: bool = True):
        got = self._exchange(int(value), range(self.world))

        class _R:
            def result(self_inner):
                return [got[r] for r in range(self.world)]
        return _R()

    def all_gather_object(self, obj) -> list:
        got = self._exchange(obj, range(self.world))
        return [got[r] for r in range(self.world)]

    def broadcast_object(self, obj, src: int = 0):
        return self._exchange(obj, range(self.world))[src]

    def max_float(self, v: float, device=None) -> float:
        return max(self._exchange(v, range(self.world)).values())

    def barrier(self) -> None:
        self.w.barrier_of(tuple(range(self.world))).wait()
