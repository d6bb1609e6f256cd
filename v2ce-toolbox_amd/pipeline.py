"""The driver loop of the hot path: frames in host memory -> packed events in host memory.

Replaces the serial structure of ``/root/reference/v2ce.py:131-209,338-367`` (the whole clip
through the model, voxel grids to host numpy and back, LDATI over the whole clip, per-frame D2H,
``np.concatenate``) by a per-batch pipeline on three HIP streams:

* copy-in stream : u8 frames of batch k+1, pinned staging -> HBM (4x fewer PCIe bytes than f32 pairs),
* main stream    : preprocess (``v2ce_preprocess_pairs``) -> V2ce3d -> LDATI count of batch k; the
                   emit phase of batch k-1 is enqueued BEHIND the model of batch k, so the one host
                   synchronisation LDATI needs (reading the segment table) never drains the GPU,
* copy-out stream: packed 13-byte records of batch k-1 -> one growing pinned host buffer, which is
                   returned as the structured array (no host-side concatenation) -- or, with a ``writer``, a ring of
                   pinned staging buffers behind which a thread writes the reference's ``.npz`` while the clip runs
                   (``StreamingEventSink`` + ``npz_stream.NpzStreamWriter``; the CLI's default).

The counter-based Philox draws make LDATI independent of the chunking, so a batch of sequences is
also the LDATI chunk (``rng='torch'`` keeps the reference's --stage2_batch_size chunks over the whole
clip: ``v2ce.events_from_voxels``).

Multi-GPU (one process per GPU, ``torch.distributed``; ``dist.py``):

* every rank walks ALL batches in lockstep and takes a contiguous share of each batch's SEQUENCES (BASELINE config
  3: ``-b 32`` on 8 GPUs = four sequences per GPU per model call); no data-path collective, the spectral-norm
  iteration of call k happens on every rank at step k, and a sequence's result does not depend on the batch it is
  launched in (per-element range slots);
* pano with a world that is a multiple of the tile count (BASELINE config 4): rank g of a tile group runs tile g
  (reference call index batch*tiles + g), then one all-to-all re-shards W-tiles -> frame-pairs so every rank
  runs full-width LDATI on its pairs; sequences are shared out over the groups;
* the packed records stream to rank 0 batch by batch (``dist.StreamedGather`` on a communication stream; rank
  order inside a step is frame-pair order), straight into rank 0's sink (host array or streamed file).
"""
from __future__ import annotations

import collections
import contextlib
import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import dist as vdist
from . import glue

EVENT_BYTES = 13


@dataclass
class BatchPlan:
    index: int                 # batch index: the reference makes calls_per_batch model calls for it
    seqs: List[int]            # sequence indices
    starts: List[int]          # first frame of each sequence
    drop: int                  # frame-pairs dropped at the start of the LAST sequence (v2ce.py:229-232)
    first_pair: int            # global index of the first kept frame-pair
    n_pairs: int               # kept frame-pairs


def plan_batches(frame_count: int, seq_len: int, batch_size: int) -> List[BatchPlan]:
    """v2ce.py:149-154,179-190,211-239 as a plan: batches of `batch_size` sequences; of the overlapped
    last sequence only the last `mode` pairs are kept."""
    sequence_num, mode, starts = glue.sequence_plan(frame_count, seq_len)
    plans = []
    for bi, s0 in enumerate(range(0, sequence_num, batch_size)):
        seqs = list(range(s0, min(s0 + batch_size, sequence_num)))
        drop = seq_len - mode if (mode != 0 and seqs[-1] == sequence_num - 1) else 0
        plans.append(BatchPlan(bi, seqs, [int(starts[s]) for s in seqs], drop, s0 * seq_len,
                               len(seqs) * seq_len - drop))
    return plans


def pano_tiles(full_width: int, width: int):
    """v2ce.py:103-111,121-122: (lo, hi, keep) per tile; keep = trailing output columns kept (0 = all).
    The exact-division test is the reference's hard-coded ``% 346`` (v2ce.py:104)."""
    patch_num = int(np.ceil(full_width / width))
    exact_div = full_width % 346 == 0
    rem = full_width % width
    tiles = []
    for i in range(patch_num):
        if i == patch_num - 1 and not exact_div:
            tiles.append((full_width - width, full_width, rem))
        else:
            tiles.append((i * width, (i + 1) * width, 0))
    return tiles


def resized_width(frames: np.ndarray, height: int) -> int:
    """v2ce.py:57: width after cv2.resize to `height` rows."""
    return int(frames.shape[2] / frames.shape[1] * height)


_OUT_CACHE = {"buf": None}     # page-locked output buffer kept between clips (reuse_output=True)


class EventSink:
    """Packed records of successive batches -> one pinned host buffer, filled by asynchronous D2H
    copies on a copy stream; ``result()`` returns it as the structured array.

    Page-locking the buffer is the slow part of a short clip (~70 us per MB, i.e. ~140 ms for the
    1.9 GB of 640 frame-pairs, against ~55 GB/s of D2H): with ``reuse=True`` the buffer is kept in the
    process and handed out again, so only the first clip pays for it -- the returned array then
    ALIASES that buffer and is overwritten by the next clip (copy it to keep it)."""

    PIN_LIMIT_BYTES = int(os.environ.get("V2CE_PIN_LIMIT_GB", 48)) << 30

    def __init__(self, device, total_pairs: int, to_host: bool = True, reuse: bool = False):
        self.reuse = reuse
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda" and to_host   # else: keep the device / CPU tensors
        self.total_pairs = max(int(total_pairs), 1)
        self.buf: Optional[torch.Tensor] = None
        self.used = 0
        self.pairs_seen = 0
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.inflight = collections.deque()
        self.cpu_parts: List[torch.Tensor] = []
        self.last_done = None                               # HIP event behind the last enqueued D2H copy

    def _reserve(self, nbytes: int):
        need = self.used + nbytes
        if self.buf is not None and need <= self.buf.numel():
            return
        # size from the event rate seen so far, with headroom; growth is geometric
        rate = (need / max(self.pairs_seen, 1)) if self.pairs_seen else 0
        cap = max(int(rate * self.total_pairs * 1.25) + (1 << 20), need, 2 * (self.buf.numel() if self.buf is not None else 0))
        cached = _OUT_CACHE["buf"] if self.reuse else None
        if cached is not None and cached is not self.buf and cached.numel() >= need:
            new = cached
        else:
            # page-locking is bounded: a clip whose events outgrow PIN_LIMIT_BYTES (the CLI streams to disk instead and
            # never gets here) continues in pageable memory -- slower copies, but a data-dependent size must not lock
            # hundreds of GB of host RAM
            pin = cap <= self.PIN_LIMIT_BYTES
            if not pin and not getattr(self, "_warned", False):
                self._warned = True
                import logging
                logging.getLogger("V2CE").warning(f"event buffer of {cap / 2**30:.0f} GiB: beyond the page-locking limit, using pageable memory")
            new = torch.empty(cap, dtype=torch.uint8, pin_memory=pin)
            if self.reuse:
                _OUT_CACHE["buf"] = new
        if self.buf is not None and self.used:
            self.stream.synchronize()
            new[:self.used].copy_(self.buf[:self.used])
        self.buf = new

    def push(self, packed: torch.Tensor, n_pairs: int, keep=(), src_stream=None):
        """`packed` was produced on `src_stream` (default: the current stream of its device)."""
        n = int(packed.numel())
        self.pairs_seen += n_pairs
        if not self.cuda:
            self.cpu_parts.append(packed)
            return
        self.pairs_seen = max(self.pairs_seen, 1)
        self._reserve(n)
        if n:
            ready = torch.cuda.Event()
            ready.record(src_stream if src_stream is not None else torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.buf[self.used:self.used + n].copy_(packed, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.stream)
            self.inflight.append((done, packed, keep))
            self.last_done = done
        self.used += n
        while self.inflight and self.inflight[0][0].query():
            self.inflight.popleft()

    def tensor(self) -> torch.Tensor:
        """The pushed buffers back to back, where they live (multi-GPU: on the device, for the gather)."""
        assert not self.cuda
        return torch.cat(self.cpu_parts) if self.cpu_parts else torch.empty(0, dtype=torch.uint8, device=self.device)

    def result(self, dtype) -> np.ndarray:
        if not self.cuda:
            return np.ascontiguousarray(self.tensor().cpu().numpy()).view(dtype)
        if self.buf is None:
            return np.empty(0, dtype)
        self.stream.synchronize()
        self.inflight.clear()
        return self.buf[:self.used].numpy().view(dtype)


class StreamingEventSink:
    """The ``EventSink`` interface in front of a file: packed records of successive batches -> a small ring of pinned
    staging buffers (asynchronous D2H on a copy stream) -> ``writer.write`` on a writer thread, in push order
    (``npz_stream.NpzStreamWriter``: the reference's ``np.savez(..., event_stream=...)`` file, v2ce.py:371-372, written
    while the clip is still running).  The host holds at most `slots` batches; ``push`` blocks when the disk falls that
    far behind.  ``result()`` drains, closes the writer and returns None (the events are in the file)."""

    def __init__(self, device, writer, slots: int = 3):
        import queue
        import threading
        self.device, self.writer = torch.device(device), writer
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.free = queue.Queue()
        for _ in range(slots):
            self.free.put([None])                          # a slot = [pinned uint8 tensor or None]
        self.work = queue.Queue()
        self.error = None
        self.last_done = None
        self.thread = threading.Thread(target=self._drain, daemon=True)
        self.thread.start()

    def _drain(self):
        while True:
            item = self.work.get()
            if item is None:
                return
            slot, n, done, _keep = item
            try:
                if self.error is None:
                    if done is not None:
                        done.synchronize()
                    self.writer.write(slot[0][:n].numpy() if done is not None else slot[0])
            except BaseException as e:                     # noqa: BLE001 -- re-raised by push / result on the caller's thread
                self.error = e
            finally:
                if done is not None:
                    self.free.put(slot)

    def push(self, packed: torch.Tensor, n_pairs: int, keep=(), src_stream=None):
        if self.error is not None:
            raise self.error
        n = int(packed.numel())
        if not n:
            return
        if not packed.is_cuda:                             # CPU stand-ins: straight to the writer thread, no staging
            self.work.put(([packed.numpy()], n, None, None))
            return
        slot = self.free.get()                             # blocks while every staging buffer waits for the disk
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(int(n * 1.25) + (1 << 20), dtype=torch.uint8, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(src_stream if src_stream is not None else torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            slot[0][:n].copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.last_done = done
        self.work.put((slot, n, done, (packed, keep)))    # the device buffers stay referenced until the copy has landed

    def result(self, dtype):
        self.work.put(None)
        self.thread.join()
        self.thread = None
        if self.error is not None:
            raise self.error
        self.writer.close()
        return None

    def abort(self):
        """Error path: stop the writer thread and drop the partial file."""
        if self.thread is not None:
            self.error = self.error or RuntimeError("aborted")
            self.work.put(None)
            self.thread.join()
            self.thread = None
        self.writer.abort()


class FrameFeeder:
    """u8 frames of one batch: pageable numpy -> pinned staging (two slots) -> HBM on a copy stream."""

    def __init__(self, frames: np.ndarray, seq_len: int, device, height: int):
        self.frames, self.seq_len, self.device = frames, seq_len, torch.device(device)
        self.cuda = self.device.type == "cuda"
        # u8 frames go to the device as they are (4x fewer PCIe bytes); resize + normalisation happen there
        self.device_pre = self.cuda and frames.dtype == np.uint8
        self.height = height
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.slots = [None, None]
        self.slot_events = [None, None]
        self.n = 0

    def submit(self, bp: BatchPlan):
        """Start moving the batch's frames; returns a handle for ``take``."""
        if not self.device_pre:
            units = np.stack([glue.image_pre_processing(self.frames[s:s + self.seq_len + 1], self.height)
                              for s in bp.starts])
            t = torch.from_numpy(units)
            return ("units", t.to(self.device, non_blocking=True) if self.cuda else t, None)
        k = self.n % 2
        self.n += 1
        b, L1 = len(bp.starts), self.seq_len + 1
        shape = (b * L1,) + self.frames.shape[1:]
        if self.slots[k] is None or self.slots[k].shape[0] < shape[0]:
            self.slots[k] = torch.empty((max(shape[0], 1),) + shape[1:], dtype=torch.uint8, pin_memory=True)
        if self.slot_events[k] is not None:
            self.slot_events[k].synchronize()               # the H2D that last read this slot
        host = self.slots[k][:shape[0]]
        hv = host.numpy()
        for i, s in enumerate(bp.starts):
            hv[i * L1:(i + 1) * L1] = self.frames[s:s + L1]
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.slot_events[k] = ev
        return ("u8", dev, ev)

    def take(self, handle, bp: BatchPlan) -> torch.Tensor:
        """[b,16,2,H,W'] f32 units on the device (main stream)."""
        kind, t, ev = handle
        if kind == "units":
            return t
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        t.record_stream(main)
        b, L1 = len(bp.starts), self.seq_len + 1
        units = [glue.image_pre_processing_device(t[i * L1:(i + 1) * L1], self.height) for i in range(b)]
        return torch.stack(units) if b > 1 else units[0][None]


def _voxels_of_batch(pred: torch.Tensor, bp: BatchPlan, seq_len: int) -> torch.Tensor:
    """[b,16,20,H,W] -> kept frame-pairs [n_pairs,2,10,H,W] (v2ce.py:211-239 for this batch)."""
    H, W = pred.shape[-2:]
    vox = pred.reshape(-1, 2, 10, H, W)
    if bp.drop:
        keep = vox.shape[0] - seq_len
        vox = torch.cat([vox[:keep], vox[keep + bp.drop:]])
    return vox


def default_stage2(fps, seed, total_pairs, device):
    """LDATI on the HIP device, split in begin (count, enqueued now) / finish (emit, later).  The
    per-frame offsets int(i*1/fps*1e6) (v2ce.py:365) of the whole clip go to the device once."""
    from .LDATI import ldati_begin
    offsets = torch.tensor([glue.frame_offset_us(i, fps) for i in range(total_pairs)], dtype=torch.int64).to(device)

    def begin(vox, first_pair):
        add = offsets[first_pair:first_pair + vox.shape[0]]
        return ldati_begin(vox, fps=fps, seed=seed, frame_base=first_pair, frame_ts_add=add)

    def finish(pending):
        ev = pending.finish()
        return ev.packed(), ev
    return begin, finish


def event_frame_sums(vox: torch.Tensor) -> torch.Tensor:
    """[P,2,10,H,W] -> [P,3,H,W]: the per-polarity sums over the ten bins (np.sum(axis=2), v2ce.py:255)
    and the sum over all twenty planes (np.sum(axis=(1,2)), v2ce.py:259), each accumulated in f32 in
    the plane order numpy uses, so the uint8 frames derived from them equal the reference's."""
    planes = vox.reshape(vox.shape[0], 20, *vox.shape[3:])
    out = torch.empty((vox.shape[0], 3, *vox.shape[3:]), dtype=torch.float32, device=vox.device)
    for c, idx in enumerate((range(0, 10), range(10, 20), range(0, 20))):
        acc = planes[:, idx[0]]
        for j in idx[1:]:
            acc = acc + planes[:, j]
        out[:, c] = acc
    return out


def shard_of_batch(bp: BatchPlan, seq_len: int, part: int, parts: int) -> BatchPlan:
    """The contiguous share `part` of `parts` of a batch's sequences as a plan of its own (may be empty):
    only the batch's LAST sequence is the overlapped one, so only the share that holds it drops pairs."""
    lo, hi = vdist.shard_range(len(bp.seqs), part, parts)
    drop = bp.drop if (hi == len(bp.seqs) and hi > lo) else 0
    return BatchPlan(bp.index, bp.seqs[lo:hi], bp.starts[lo:hi], drop, bp.first_pair + lo * seq_len,
                     (hi - lo) * seq_len - drop)


def _shared_segment_path() -> str:
    """A fresh file name for the shared host segment of one clip (tmpfs: its pages are host memory)."""
    import uuid
    base = "/dev/shm" if os.path.isdir("/dev/shm") else (os.environ.get("TMPDIR") or "/tmp")
    return os.path.join(base, f"v2ce_events_{os.getpid()}_{uuid.uuid4().hex}.bin")


def _registered_window_bytes(n_pairs: int, device) -> int:
    """Bytes of the shared host segment that every rank page-locks and fills by DMA (dist.HostDirectGather): V2CE_HOST_SEGMENT_MB,
    default min(2 MiB per frame-pair -- 4.7 Mevents: twice what UNet output at 346x260 yields --, half of what /dev/shm has free);
    0 on CPU runs unless the variable asks for one (the same window as a plain shared mapping).  A clip that outgrows the window keeps
    working: the rest takes the staging + pwrite path."""
    env = os.environ.get("V2CE_HOST_SEGMENT_MB")
    if env is not None:
        return max(0, int(env)) << 20
    if torch.device(device).type != "cuda":
        return 0
    want = int(n_pairs) * (2 << 20)
    try:
        st = os.statvfs("/dev/shm" if os.path.isdir("/dev/shm") else (os.environ.get("TMPDIR") or "/tmp"))
        want = min(want, st.f_bavail * st.f_frsize // 2)
    except OSError:
        pass
    return max(0, want) & ~4095


def _host_identity() -> str:
    """Something all ranks of one machine share and ranks of different machines do not (gather='host' precondition)."""
    import socket
    boot = ""
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        pass
    return socket.gethostname() + ":" + boot


def run_clip(frames: np.ndarray, model, *, infer_type="center", seq_len=16, width=346, height=260,
             batch_size=1, fps=30, seed=0, device="cuda", stage2=None, dtype=None,
             comm=None, trace: Optional[dict] = None,
             reuse_output: bool = False, event_frames: Optional[list] = None, writer=None,
             gather: Optional[str] = None) -> Optional[np.ndarray]:
    """frames [N,H,W] uint8 -> event_stream (structured array) on rank 0, None elsewhere.

    writer: an ``npz_stream.NpzStreamWriter`` (rank 0): the records go to the file batch by batch instead of into one
    host array (``StreamingEventSink``); the function then returns None on every rank and closes the writer.

    comm: ``dist.TorchComm`` / ``dist.ThreadComm`` / ``dist.LocalComm`` (default: from torch.distributed).
    gather (more than one rank; V2CE_GATHER overrides; default ``dist.default_gather_mode``: 'device' at every world size -- 'host'
    is opt-in until a run on an 8-GPU node has shown both): 'device' -- the records are gathered on rank 0's GPU over RCCL / xGMI (``dist.StreamedGather``) and rank 0
    downloads them into its pinned sink (one PCIe link: ~57 GB/s measured, against 8 x 7.3 GB/s of records at N = 8 in the e2e
    regime: the link is the budget there); 'host' -- every rank brings its own records to host memory over its own
    PCIe link, into its slice of the output (``dist.HostDirectGather``; only byte counts cross RCCL): a shared tmpfs segment
    whose first V2CE_HOST_SEGMENT_MB every rank page-locks and fills by GPU DMA (round 5: faster than the device gather even
    on one GPU), the rest -- and the streamed .npz, whose CRC needs the CPU anyway -- through pinned staging and pwrite
    (~6 GB/s per inode, tools/shm_write_probe.py).
    A rank that fails inside the clip makes EVERY rank raise at the same step (``dist.RankFailure``; the failed rank
    re-raises its own error), and the LDATI status words are reduced over the ranks before rank 0 finalises its output.
    stage2: optional (begin, finish) pair replacing LDATI (CPU stand-ins in the tests):
    begin(vox, first_pair) -> handle; finish(handle) -> (packed uint8 tensor, keepalive).
    event_frames: a list that receives, per batch, (first pair, event_frame_sums(voxels)) on the device
    -- the reduction the reference's event-frame video starts from (v2ce.py:254-260); single-process
    runs only."""
    from .LDATI import EVENT_DTYPE
    dtype = dtype or EVENT_DTYPE
    comm = comm or vdist.default_comm()
    rank, world = comm.rank, comm.world
    if infer_type not in ("center", "pano"):
        raise ValueError(f"Invalid infer_type {infer_type}")
    gather = gather or vdist.default_gather_mode(world)
    if gather not in ("host", "device"):
        raise ValueError(f"gather must be 'host' or 'device', got {gather!r}")
    plans = plan_batches(len(frames), seq_len, batch_size)
    begin0, finish0 = stage2 or default_stage2(fps, seed, len(frames) - 1, device)
    # Stage 2 runs on its OWN stream beside the next batch's convolutions (round 6).  The conv kernels are persistent -- one workgroup per
    # CU, which its LDS and registers fill -- so LDATI's short workgroups run where a conv launch has no tile left for a CU: its
    # 0.4 ms per batch disappear in the tails of the thirty launches of a forward call (16.38 -> 16.03 ms per 64 frame-pairs).
    # V2CE_LDATI_STREAM=0: behind the model on the main stream, as before.
    ldati_stream = None
    if torch.device(device).type == "cuda" and os.environ.get("V2CE_LDATI_STREAM", "1") != "0":
        ldati_stream = torch.cuda.Stream(device=device, priority=int(os.environ.get("V2CE_LDATI_STREAM_PRIORITY", "-1")))

    def begin(vox, first_pair):
        if ldati_stream is None:
            return begin0(vox, first_pair)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(device))     # the model's output (and anything else the main stream did to vox)
        with torch.cuda.stream(ldati_stream):
            ldati_stream.wait_event(ready)
            handle = begin0(vox, first_pair)
        vox.record_stream(ldati_stream)                     # (the caching allocator must not hand vox out again before stage 2 is done)
        return handle

    def finish(handle):
        if ldati_stream is None:
            return finish0(handle)
        with torch.cuda.stream(ldati_stream):
            return finish0(handle)
    fw = resized_width(frames, height)
    tiles = pano_tiles(fw, width) if infer_type == "pano" else None
    calls_per_batch = len(tiles) if tiles else 1
    tile_parallel = bool(tiles) and len(tiles) > 1 and world > 1 and world % len(tiles) == 0
    if tile_parallel:
        # groups of len(tiles) ranks: rank g of a group runs tile g; the batch's sequences are shared out over the groups
        seq_part, tile_index = divmod(rank, len(tiles))
        seq_parts = world // len(tiles)
        grp = comm.tile_group(len(tiles))
        widths = [(k if k else width) for _, _, k in tiles]
    else:
        seq_part, seq_parts, tile_index, grp = rank, world, None, None
    mine = [shard_of_batch(bp, seq_len, seq_part, seq_parts) for bp in plans]
    feeder = FrameFeeder(frames, seq_len, device, height)
    multi = not isinstance(comm, vdist.LocalComm)           # (a forced world of one takes the collective path too)
    host_direct = multi and gather == "host"
    if host_direct:
        # 'host' needs every rank on rank 0's machine (one tmpfs / one local file system): checked, not assumed
        ids = comm.all_gather_object(_host_identity())
        if len(set(ids)) > 1:
            import logging
            logging.getLogger("V2CE").warning("gather='host' needs all ranks on one host (%d distinct hosts): using 'device'", len(set(ids)))
            host_direct, gather = False, "device"
    # single rank, or device gather: rank 0 owns a sink that downloads everything
    sink = None
    if rank == 0 and not host_direct:
        sink = StreamingEventSink(device, writer) if writer is not None else EventSink(device, len(frames) - 1, reuse=reuse_output)
    exchange, seg_path = None, None
    if host_direct:
        # the shared output: rank 0's streamed .npz (its records start behind the headers), or a host segment in tmpfs
        if rank == 0:
            if writer is not None:
                writer.f.flush()
                info = (writer.part_path, writer.records_start, True)
            else:
                seg_path = _shared_segment_path()
                # the registered window (V2CE_HOST_SEGMENT_MB; 0 = none): sized here, page-locked by every rank, written by DMA
                reg = _registered_window_bytes(len(frames) - 1, torch.device(device))
                with open(seg_path, "wb") as f:
                    if reg:
                        f.truncate(reg)
                info = (seg_path, 0, False, reg)
        else:
            info = None
        info = comm.broadcast_object(info, src=0)
        path, data_start, need_crc = info[:3]
        try:
            exchange = vdist.HostDirectGather(comm, device, path, data_start, need_crc, registered_bytes=info[3] if len(info) > 3 else 0)
        except BaseException:
            # the constructor is collective and raises on every rank when one of them cannot open or page-lock the file: rank 0
            # takes the segment (and a half-written .npz) with it
            if rank == 0:
                if seg_path is not None and os.path.exists(seg_path):
                    os.unlink(seg_path)
                if writer is not None:
                    writer.abort()
            raise
    elif multi:
        step_pairs = collections.deque(bp.n_pairs for bp in plans)

        def on_pieces(pieces, stream):
            n = step_pairs.popleft()
            for k, p in enumerate(pieces):                  # rank order = frame-pair order inside a step
                # (CPU stand-ins keep the tensors they are handed: the receive buffers are reused, so they get copies)
                sink.push(p if stream is not None else p.clone(), n if k == 0 else 0, src_stream=stream)
            return sink.last_done if stream is not None else None    # the receive buffers are reused behind these copies
        exchange = comm.streamed_gather(on_pieces if rank == 0 else None, dst=0)
    base_calls = int(getattr(model, "calls", 0))            # the reference keeps advancing one model
    import time as _time

    def tick(name, t0):
        if trace is not None:
            trace[name] = trace.get(name, 0.0) + _time.perf_counter() - t0
            trace.setdefault("list:" + name, []).append(round(1e3 * (_time.perf_counter() - t0), 2))
        return _time.perf_counter()

    status = {"acc": None, "msg": None}                     # device-side OR of the LDATI status words of the clip
    failure = {"exc": None}                                 # this rank's first error (multi-rank: reported with the next byte count)
    pending = None                                          # (stage-2 handle, pairs)

    def guarded(fn, *args):
        """Multi-rank: an error is kept and travels to the peers with the next exchange (every rank then stops at the
        same step); single rank: it propagates at once."""
        if failure["exc"] is not None:
            return None
        if not multi:
            return fn(*args)
        try:
            return fn(*args)
        except vdist.RankFailure:
            raise
        except Exception as e:                              # noqa: BLE001 -- re-raised below, on every rank
            failure["exc"] = e
            return None

    def flush(p):
        handle, n_pairs = p
        t_f = _time.perf_counter()
        packed, keep = torch.empty(0, dtype=torch.uint8, device=device), None
        if handle is not None:                              # (else: this rank had no pairs in that batch)
            got = guarded(finish, handle)
            if got is not None:
                packed, keep = got
        t_f = tick("flush.finish", t_f)
        st = getattr(keep, "_status", None)
        if st is not None:                                  # 4 bytes folded on the stream; the event object is not retained
            with torch.cuda.stream(ldati_stream) if ldati_stream is not None else contextlib.nullcontext():
                if status["acc"] is None:
                    status["acc"] = torch.zeros_like(st)
                torch.maximum(status["acc"], st, out=status["acc"])
            status["msg"] = getattr(keep, "_status_message", None) or status["msg"]
        if ldati_stream is not None and multi and handle is not None:
            # the exchanges read the records on the main stream: behind stage 2 (which ran beside the model call enqueued since)
            emitted = torch.cuda.Event()
            emitted.record(ldati_stream)
            torch.cuda.current_stream(device).wait_event(emitted)
            packed.record_stream(torch.cuda.current_stream(device))
        if host_direct:
            exchange.submit(packed, failed=failure["exc"] is not None, keep=keep)
        elif multi:
            exchange.submit(packed, failed=failure["exc"] is not None)
        else:
            sink.push(packed, n_pairs, keep, src_stream=ldati_stream if handle is not None else None)
        tick("flush.push", t_f)

    def next_nonempty(i):
        while i < len(mine) and not mine[i].seqs:
            i += 1
        return i

    def model_part(i, bp, state):
        """Everything of a batch in front of the tile exchange: this rank's voxels ([P,2,10,H,W], or its W-tile of them)."""
        units = feeder.take(state["handle"], bp)
        nxt_i = next_nonempty(i + 1)
        state["handle"] = feeder.submit(mine[nxt_i]) if nxt_i < len(mine) else None
        # call index of this batch in the reference's schedule (one spectral-norm iteration per call);
        # ranks that sat a batch out catch up here
        vdist.fast_forward(model, base_calls + bp.index * calls_per_batch + (tile_index or 0))
        if not tile_parallel:
            if infer_type == "center":
                pred = glue.infer_center_image_unit(model, units, width)
            else:
                pred = glue.infer_pano_image_unit(model, units, width)
            return _voxels_of_batch(pred, bp, seq_len)
        lo, hi, keep_cols = tiles[tile_index]
        pred = model(units[..., lo:hi].float().contiguous())
        if keep_cols:
            pred = pred[..., -keep_cols:]
        return _voxels_of_batch(pred, bp, seq_len)                                  # [P,2,10,H,wt]

    def batch_body(i, bp, state):
        vox, first_pair = None, bp.first_pair
        if bp.seqs:
            if not tile_parallel:
                vox = model_part(i, bp, state)
            else:
                # The tile exchange is a collective of the group: a rank that has failed (now or in an earlier batch) still
                # takes part, with zeros of the right shape, so that its peers are never left inside an unmatched all_to_all
                # (ADVICE r4); its failure travels with the next byte count like any other
                part = guarded(model_part, i, bp, state)
                if part is None:
                    part = torch.zeros((bp.n_pairs, 2, 10, height, widths[tile_index]), dtype=torch.float32, device=device)
                vox, p_lo = comm.tiles_to_pairs(part, widths, tile_index, grp)       # [P_r,2,10,H,W_full]
                first_pair += p_lo
        if event_frames is not None and vox is not None:
            event_frames.append((first_pair, event_frame_sums(vox)))
        if tile_parallel:
            handle = guarded(begin, vox, first_pair) if vox is not None and vox.shape[0] else None
            return (handle, 0 if vox is None else int(vox.shape[0]))
        return (begin(vox, first_pair) if vox is not None and vox.shape[0] else None, 0 if vox is None else int(vox.shape[0]))

    def cleanup_on_error():
        if host_direct:
            exchange.close()
            if seg_path is not None and os.path.exists(seg_path):
                os.unlink(seg_path)
        if writer is not None and rank == 0:
            if isinstance(sink, StreamingEventSink):
                sink.abort()
            else:
                writer.abort()

    try:
        nxt_i = next_nonempty(0)
        state = {"handle": feeder.submit(mine[nxt_i]) if nxt_i < len(mine) else None}
        with torch.no_grad():
            for i, bp in enumerate(mine):
                tt = _time.perf_counter()
                # (tile-parallel: the body guards its own parts and never skips the group's collective)
                nxt = (batch_body(i, bp, state) if tile_parallel else guarded(batch_body, i, bp, state)) or (None, 0)
                tt = tick("model+begin", tt)
                if pending is not None:
                    flush(pending)
                tt = tick("flush", tt)
                pending = nxt
            if pending is not None:
                flush(pending)
            # every rank leaves the model where the single-process run leaves it
            if failure["exc"] is None:
                vdist.fast_forward(model, base_calls + len(plans) * calls_per_batch)
        if exchange is not None:
            exchange.drain()
        # the LDATI status words of all ranks, before anything is finalised (a rank with a bad segment must not leave rank 0
        # closing a valid-looking file)
        if ldati_stream is not None:
            ldati_stream.synchronize()
        st_local = float(int(status["acc"].item())) if status["acc"] is not None else 0.0
        st_all = comm.max_float(st_local, device=device) if multi else st_local
        if st_all != 0.0:
            from . import hip
            raise hip.V2ceHipError(status["msg"] or "LDATI: a (frame, bin) segment could not be ordered on the device "
                                   "(more equal-time events than the LDS sort holds at an fps beyond the sweep kernel's histogram)"
                                   + ("" if st_local else " [reported by another rank]"))
        if host_direct:
            total = exchange.finalize()                     # collective; rank 0: (bytes, crc)
            if rank != 0:
                return None
            nbytes, crc = total
            if writer is not None:
                writer.set_external(nbytes, crc)
                writer.close()
                return None
            if nbytes == 0:
                os.unlink(seg_path)
                return np.empty(0, dtype)
            if os.path.getsize(seg_path) > nbytes:
                # the registered window was sized ahead of the clip (~2x the real volume): every rank has unregistered and closed it by
                # now, so the tail goes back to /dev/shm before the array pins the file's pages for its lifetime
                os.truncate(seg_path, nbytes)
            mm = np.memmap(seg_path, dtype=np.uint8, mode="r+", shape=(nbytes,))
            os.unlink(seg_path)                             # the mapping keeps the pages; the name goes now
            return mm.view(dtype)
        return sink.result(dtype) if rank == 0 else None
    except vdist.RankFailure as rf:
        cleanup_on_error()
        if failure["exc"] is not None:                      # this rank is (one of) the failed ones: its own error
            raise failure["exc"] from rf
        raise
    except BaseException:
        cleanup_on_error()
        raise
