"""LDATI (stage 2) -- host side mirroring ``/root/reference/scripts/LDATI.py``.

``sample_voxel_statistical`` keeps the reference signature (``scripts/LDATI.py:126``) and return
type (list of packed numpy recarrays, ``scripts/LDATI.py:308``); everything is computed by the HIP
kernels of ``csrc/ldati.hip`` through the C ABI (``include/v2ce_hip.h``).  No CPU path exists.

All option values of the reference signature are implemented (bidirectional relocation, pooled
slope, 'none' / 'random' strategies).  Differences from the reference, all deliberate:

* Random draws.  The reference draws one dense ``torch.rand([B,2,9,H,W,max_n])`` (LDATI.py:171).
  ``rng='torch'`` does exactly that on the tensor's device and replays it (same consumption of the
  global torch generator, O(voxels*max_n) memory).  The default ``rng='philox'`` draws only the
  uniforms that are used from a counter-based Philox4x32-10 keyed by ``seed`` and indexed by
  (global frame, polarity, bin, pixel, draw), so results do not depend on batching or sharding.
  ``uniforms=`` replays a caller-supplied dense tensor (how parity with the oracle is tested).
* Tie order inside a (frame, bin) segment is always the stable order (what the reference's CPU
  ``argsort`` gives for segments >= 32768 events; unstable below that).
* All-zero input: the reference raises from a debug-log reduction over an empty tensor
  (LDATI.py:200); here it returns empty recarrays unless ``strict_reference_errors=True``.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import List, Optional

import numpy as np
import torch

from . import hip

EVENT_DTYPE = np.dtype([("timestamp", "<i8"), ("x", "<i2"), ("y", "<i2"), ("polarity", "i1")])
assert EVENT_DTYPE.itemsize == 13


class DeviceEvents:
    """Events of one LDATI call, resident on the device as packed 13-byte records (the numpy record
    layout of LDATI.py:308) + the host-side segment table.  ``ts`` / ``x`` / ``y`` / ``p`` are SoA
    views materialised on first use (``v2ce_events_unpack``)."""

    def __init__(self, packed, seg_counts, max_n, soa=None):
        self._packed = packed                   # uint8 [N*13] device tensor (or None when soa is given)
        self._soa = soa                         # (ts i64, x i16, y i16, p i8) device tensors
        self.seg_counts = seg_counts            # np.int64 [B, 9]
        self.max_n = max_n
        self._status = None                     # device int32[1]: see check()

    @property
    def num_events(self) -> int:
        return int(self.seg_counts.sum())

    @property
    def frame_counts(self) -> np.ndarray:
        return self.seg_counts.sum(axis=1)

    @property
    def device(self):
        return (self._packed if self._packed is not None else self._soa[0]).device

    def _unpacked(self):
        if self._soa is None:
            n, dev = self.num_events, self._packed.device
            ts = torch.empty(n, dtype=torch.int64, device=dev)
            x = torch.empty(n, dtype=torch.int16, device=dev)
            y = torch.empty(n, dtype=torch.int16, device=dev)
            p = torch.empty(n, dtype=torch.int8, device=dev)
            if n:
                with torch.cuda.device(dev):
                    hip.check(hip.lib().v2ce_events_unpack(self._packed.data_ptr(), n, ts.data_ptr(), x.data_ptr(),
                                                           y.data_ptr(), p.data_ptr(), hip.stream_ptr(dev)),
                              "v2ce_events_unpack")
            self._soa = (ts, x, y, p)
        return self._soa

    ts = property(lambda self: self._unpacked()[0])
    x = property(lambda self: self._unpacked()[1])
    y = property(lambda self: self._unpacked()[2])
    p = property(lambda self: self._unpacked()[3])

    def packed(self) -> torch.Tensor:
        """uint8 [N*13] device tensor of packed records (LDATI.py:308 dtype)."""
        if self._packed is None:
            ts, x, y, p = self._soa
            n = self.num_events
            out = torch.empty(max(n * 13, 4), dtype=torch.uint8, device=ts.device)
            if n:
                with torch.cuda.device(ts.device):
                    hip.check(hip.lib().v2ce_events_pack(ts.data_ptr(), x.data_ptr(), y.data_ptr(), p.data_ptr(), n,
                                                         out.data_ptr(), hip.stream_ptr(ts.device)), "v2ce_events_pack")
            self._packed = out[: n * 13]
        return self._packed

    def check(self) -> None:
        """Raise if the device reported a segment it could not order (a coarse bucket beyond the LDS
        capacity at an fps whose key range the sweep fallback cannot hold).  Synchronises."""
        if self._status is not None and int(self._status.item()) != 0:
            raise hip.V2ceHipError(getattr(self, "_status_message", None) or
                                   "LDATI: a (frame, bin) segment has more equal-time events than the LDS sort "
                                   "holds and the key range of this fps exceeds the sweep kernel's histogram")

    def to_recarrays(self) -> List[np.recarray]:
        ev = np.ascontiguousarray(self.packed().cpu().numpy()).view(EVENT_DTYPE)
        self.check()
        out, lo = [], 0
        for n in self.frame_counts:
            out.append(ev[lo:lo + int(n)].copy().view(np.recarray))
            lo += int(n)
        return out


def _check_fps(fps) -> None:
    # LDATI.py:163 reshapes arange(0, 1/fps, 1/fps/9) to 9 bins; other lengths raise there
    frame_step = 1 / fps
    voxel_step = 1 / fps / 9
    if math.ceil(frame_step / voxel_step) != 9:
        raise RuntimeError(f"shape '[1, 1, 9, 1, 1]' is invalid: arange(0, 1/{fps}, 1/{fps}/9) "
                           f"does not have 9 elements (reference LDATI.py:163 raises as well)")


class _PinnedPool:
    """Small pinned int64 buffers for the segment-table read-back, recycled by hand: the allocator's
    own pinned cache cannot reuse a block while an asynchronous copy into it is pending and would call
    hipHostMalloc (which synchronises the device) on every pipelined call."""

    def __init__(self):
        self.free = {}

    def get(self, n: int) -> torch.Tensor:
        lst = self.free.setdefault(n, [])
        return lst.pop() if lst else torch.empty(n, dtype=torch.int64, pin_memory=True)

    def put(self, t: torch.Tensor) -> None:
        self.free.setdefault(int(t.numel()), []).append(t)


_PINNED = _PinnedPool()
# densest-segment event count of the previous call with the same shape and options: the fused count pass assumes the
# bucket geometry that follows from it (consecutive batches of a clip agree; a miss costs one two-pass call)
_SEG_HINT = {}
_SPARSE_TILE_CAP = 8192                  # kSparseCap of csrc/ldati.hip: events of one tile (all nine bins) the fused pass holds


class PendingLdati:
    """An LDATI call whose count phase is enqueued (``ldati_begin``).  ``finish()`` waits for the
    segment table -- the one host synchronisation inherent to a variable-length output --, allocates
    and enqueues the emit phase.  Splitting the call lets a caller enqueue other GPU work (the next
    batch's UNet) before it blocks on the counts."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def finish(self) -> DeviceEvents:
        with torch.cuda.device(self.y.device):
            return _ldati_finish(self)


def ldati_begin(y: torch.Tensor, t0=0, fps=30, *, rng: str = "philox", seed: Optional[int] = None,
                frame_base: int = 0, uniforms: Optional[torch.Tensor] = None,
                frame_ts_add: Optional[torch.Tensor] = None, profile: Optional[list] = None,
                path: str = "bucket", strategy: str = "slope", layout: str = "packed",
                bidirectional: bool = False, pooling_type: str = "none", pooling_kernel_size: int = 3) -> PendingLdati:
    """Enqueue the count phase of LDATI on y [B,2,10,H,W] (HIP device) and start the asynchronous
    copy of the segment table to pinned host memory."""
    if y.dim() != 5 or y.shape[1] != 2 or y.shape[2] != 10:
        raise ValueError(f"expected y of shape [B,2,10,H,W], got {tuple(y.shape)}")
    _check_fps(fps)
    if not y.is_cuda:
        raise hip.V2ceHipError("sample_voxel_statistical: y must be on a HIP device; no CPU path")
    if path not in ("bucket", "sweep"):
        raise ValueError(f"path must be 'bucket' or 'sweep', got {path!r}")
    if layout not in ("packed", "soa"):
        raise ValueError(f"layout must be 'packed' or 'soa', got {layout!r}")
    if rng not in ("philox", "torch"):
        raise ValueError(f"rng must be 'philox' or 'torch', got {rng!r}")
    # the C ABI launches on the current device's stream: make y's device current for the call
    with torch.cuda.device(y.device):
        y = y.float().contiguous()                     # LDATI.py:143 `.float()`
        B, _, _, H, W = y.shape
        dev = y.device
        L = hip.lib()
        st = hip.stream_ptr(dev)
        if path == "sweep" and L.v2ce_ldati_lds_bytes(float(fps), float(t0)) == 0:
            raise hip.V2ceHipError(f"fps={fps}, t0={t0}: time bin too wide for the sweep kernel's LDS key histogram")
        opts = hip.LdatiOptions(
            strategy={"slope": hip.STRATEGY_SLOPE, "none": hip.STRATEGY_NONE, "random": hip.STRATEGY_RANDOM}[strategy],
            bidirectional=int(bool(bidirectional)),
            pooling_type={"none": hip.POOL_NONE, "avg": hip.POOL_AVG, "weighted": hip.POOL_WEIGHTED}[pooling_type],
            pooling_kernel_size=int(pooling_kernel_size))
        plain = strategy != "random" and not bidirectional and (pooling_type == "none" or strategy != "slope")
        if path == "sweep" and not plain:
            raise hip.V2ceHipError("path='sweep' covers forward relocation without pooling ('slope' / 'none')")
        tile_ws = torch.empty(L.v2ce_ldati_tile_ws_bytes(B, H, W), dtype=torch.uint8, device=dev)
        meta = torch.empty(B * 9 + 1 + 8, dtype=torch.int64, device=dev)       # seg_offsets | stats
        # the random draws are fixed here (not in finish): the fused count below already computes timestamps
        mode, u_keep, replay_max_n = hip.RNG_PHILOX, None, 0
        if uniforms is not None:
            u_keep = hip.require_device_f32(uniforms.to(dev), "uniforms")
            if u_keep.dim() != 6 or tuple(u_keep.shape[:5]) != (B, 2, 9, H, W):
                raise ValueError(f"uniforms must be [B,2,9,H,W,>=max_n], got {tuple(u_keep.shape)}")
            mode, replay_max_n = hip.RNG_REPLAY, int(u_keep.shape[5])
        elif rng == "philox" and seed is None:     # reproducible under torch.manual_seed, like the reference's draw
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        # One pass over the voxels instead of two (v2ce_ldati_count_fused): counts AND the sparse tiles' records; finish()
        # falls back to the two-pass path by itself when a tile was dense.  rng='torch' draws its tensor after the counts.
        fused_ws, seg_hint, bin_hint, hint_key = None, 0, 0, None
        if path == "bucket" and (rng == "philox" or uniforms is not None) and os.environ.get("V2CE_LDATI_NO_FUSED") is None:
            hint_key = (dev.index, B, H, W, float(fps), float(t0), strategy, bool(bidirectional))
            seg_hint, tile_hint, calls, bin_last = _SEG_HINT.get(hint_key, (0, 0, 0, 0))
            # Sparse tiles last time (or no history): the sparse kernel's fused form, a slot per tile.  A dense tile last time:
            # the dense kernel's fused form, a slot per (tile, bin) sized from last time's densest one plus a margin (a miss
            # costs the pass and its two-pass repeat; the bytes are the same) -- and the sparse form again every 16th call, so
            # that a stream that turns sparse is picked up.
            fb = 0
            if tile_hint <= _SPARSE_TILE_CAP or calls % 16 == 0:
                fb = L.v2ce_ldati_fused_ws_bytes(B, H, W, float(fps), float(t0), ctypes.byref(opts), seg_hint, 0)
            elif bin_last > 0:
                bin_hint = int(bin_last * 1.06) + 128
                fb = L.v2ce_ldati_fused_ws_bytes(B, H, W, float(fps), float(t0), ctypes.byref(opts), seg_hint, bin_hint)
                if not fb:
                    bin_hint = 0                          # (no dense form for these options: the plain count)
            if fb:
                fused_ws = torch.empty(fb, dtype=torch.uint8, device=dev)
        if profile is not None:    # HIP events on the launch stream around the count kernels
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
        if fused_ws is not None:
            hip.check(L.v2ce_ldati_count_fused(y.data_ptr(), B, H, W, float(fps), float(t0), ctypes.byref(opts), mode, hip.ptr(u_keep),
                                               replay_max_n, int(seed or 0) & (2 ** 64 - 1), int(frame_base), seg_hint, bin_hint,
                                               tile_ws.data_ptr(), tile_ws.numel(), fused_ws.data_ptr(), fused_ws.numel(), meta.data_ptr(),
                                               meta[B * 9 + 1:].data_ptr(), st), "v2ce_ldati_count_fused")
        else:                                          # (stats[4..7] stay unwritten: only the fused path reads stats[4])
            hip.check(L.v2ce_ldati_count(y.data_ptr(), B, H, W, ctypes.byref(opts), tile_ws.data_ptr(), tile_ws.numel(),
                                         meta.data_ptr(), meta[B * 9 + 1:].data_ptr(), st), "v2ce_ldati_count")
        if profile is not None:
            c1.record()
            profile.append(("count", c0, c1, 0))
        host = _PINNED.get(int(meta.numel()))
        host.copy_(meta, non_blocking=True)
        ready = torch.cuda.Event()
        ready.record()
    return PendingLdati(y=y, t0=t0, fps=fps, rng=rng, seed=seed, frame_base=frame_base, uniforms=u_keep,
                        frame_ts_add=frame_ts_add, profile=profile, path=path, opts=opts, plain=plain, layout=layout,
                        tile_ws=tile_ws, meta=meta, host=host, ready=ready, fused_ws=fused_ws, seg_hint=seg_hint, bin_hint=bin_hint, hint_key=hint_key)


def _ldati_finish(q: PendingLdati) -> DeviceEvents:
    y, fps, t0 = q.y, q.fps, q.t0
    B, _, _, H, W = y.shape
    dev = y.device
    L = hip.lib()
    st = hip.stream_ptr(dev)
    q.ready.synchronize()                              # the one sync
    host = q.host.numpy().copy()
    _PINNED.put(q.host)
    offs = host[:B * 9 + 1]
    max_n, max_tile, max_seg, total, tile_all = (int(v) for v in host[B * 9 + 1:B * 9 + 6])
    segc = np.diff(offs).reshape(B, 9)
    if q.hint_key is not None:
        prev = _SEG_HINT.get(q.hint_key, (0, 0, 0, 0))
        _SEG_HINT[q.hint_key] = (max_seg, tile_all if q.fused_ws is not None else prev[1], prev[2] + 1, max_tile)

    mode, u_ptr, replay_max_n = hip.RNG_PHILOX, None, 0
    keep, seed, uniforms = None, q.seed, q.uniforms
    if uniforms is not None:
        if uniforms.shape[5] < max_n:
            raise ValueError(f"uniforms must be [B,2,9,H,W,>=max_n={max_n}], got {tuple(uniforms.shape)}")
        mode, u_ptr, replay_max_n, keep = hip.RNG_REPLAY, uniforms.data_ptr(), uniforms.shape[5], uniforms
    elif q.rng == "torch":
        keep = torch.rand([B, 2, 9, H, W, max_n], device=dev)      # LDATI.py:171
        mode, u_ptr, replay_max_n = hip.RNG_REPLAY, keep.data_ptr(), max_n

    # An unphysical voxel grid (values in the thousands: a broken checkpoint, un-normalised input) asks for billions of events;
    # the reference dies there allocating its dense [B,2,9,H,W,max_n] tensors (LDATI.py:171).  Refuse before anything of that
    # size is allocated or launched (V2CE_MAX_EVENTS overrides the 2^31 default).
    limit = int(os.environ.get("V2CE_MAX_EVENTS", 1 << 31))
    # (the device-side counters are 32-bit per tile and per segment: a voxel count this large could wrap them, so it is
    # checked first -- max_n comes from a per-voxel atomic max and cannot wrap)
    if max_n > 4096 or 2 * H * W * max(max_n, 1) >= 1 << 32 or total > limit:
        raise hip.V2ceHipError(f"LDATI: {total} events in one call of {B} frame-pairs, largest voxel count {max_n}: the voxel "
                               f"values are unphysical (or the chunk is too large: limit {limit} events, V2CE_MAX_EVENTS; "
                               "4096 events per voxel and time bin)")
    packed = soa = None
    ptrs = [None] * 5
    if q.layout == "packed":
        packed = torch.empty(max(total * 13, 4), dtype=torch.uint8, device=dev)
        ptrs[4] = packed.data_ptr()
    else:
        soa = (torch.empty(total, dtype=torch.int64, device=dev), torch.empty(total, dtype=torch.int16, device=dev),
               torch.empty(total, dtype=torch.int16, device=dev), torch.empty(total, dtype=torch.int8, device=dev))
        ptrs[:4] = [t.data_ptr() for t in soa]
    ev = DeviceEvents(None if packed is None else packed[: total * 13], segc, max_n, soa)
    frame_ts_add = q.frame_ts_add
    ws = None
    if total:
        add_ptr = None
        if frame_ts_add is not None:
            frame_ts_add = frame_ts_add.to(device=dev, dtype=torch.int64).contiguous()
            assert frame_ts_add.numel() == B
            add_ptr = frame_ts_add.data_ptr()
        ws_bytes = 0
        if q.path == "bucket":     # two-level path; "sweep" = one workgroup per segment
            ws_bytes = L.v2ce_ldati_workspace_bytes(B, H, W, float(fps), float(t0), ctypes.byref(q.opts), total, max_seg,
                                                    max_tile, int(q.layout == "packed"))
            if ws_bytes:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            elif not q.plain or L.v2ce_ldati_lds_bytes(float(fps), float(t0)) == 0:
                raise hip.V2ceHipError(f"LDATI: fps={fps}, t0={t0}, {H}x{W}, {max_tile} events in one tile-bin: outside "
                                       "both the two-level path and the sweep kernel")
        if q.profile is not None:    # HIP events on the launch stream around the emit kernels
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        args = (y.data_ptr(), B, H, W, float(fps), float(t0), ctypes.byref(q.opts), mode, u_ptr,
                int(replay_max_n), int(seed or 0) & (2 ** 64 - 1), int(q.frame_base),
                q.meta.data_ptr(), add_ptr, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4],
                total, max_seg, max_tile, q.tile_ws.data_ptr(), hip.ptr(ws), int(ws_bytes))
        if q.fused_ws is not None and ws is not None:
            hip.check(L.v2ce_ldati_emit_fused(*args, q.fused_ws.data_ptr(), q.fused_ws.numel(), tile_all, q.seg_hint, q.bin_hint, st), "v2ce_ldati_emit_fused")
        else:
            hip.check(L.v2ce_ldati_emit(*args, st), "v2ce_ldati_emit")
        if q.profile is not None:
            e1.record()
            # algorithmic bytes: 80 B per pixel read once + 13 B per event written once
            q.profile.append(("emit", e0, e1, 80 * B * H * W + 13 * total))
        if ws is not None:
            sp = ctypes.c_void_p()
            hip.check(L.v2ce_ldati_status(ws.data_ptr(), B, H, W, float(fps), float(t0), ctypes.byref(q.opts), total, max_seg,
                                          max_tile, ctypes.byref(sp)), "v2ce_ldati_status")
            off = (sp.value - ws.data_ptr())
            ev._status = ws[off:off + 4].view(torch.int32)
    ev._keepalive = (y, keep, frame_ts_add, q.meta, q.tile_ws, ws, q.fused_ws)
    return ev


def ldati_device(y: torch.Tensor, t0=0, fps=30, **kw) -> DeviceEvents:
    """Run count -> emit on the device and leave the events there (``ldati_begin(...).finish()``).

    y: [B,2,10,H,W] on a HIP device.  One host synchronisation (reading the B*9+1 segment offsets
    and the allocation statistics) is inherent to the variable-length output."""
    return ldati_begin(y, t0, fps, **kw).finish()


def sample_voxel_statistical(y, t0=0, fps=30, pooling_type="none", pooling_kernel_size=3,
                             additional_events_strategy="slope", bidirectional=False, *,
                             rng: str = "philox", seed: Optional[int] = None, frame_base: int = 0,
                             uniforms=None, strict_reference_errors: bool = False):
    """Drop-in for ``scripts/LDATI.py:126``: y [B,2,10,H,W] -> list[B] of packed recarrays
    ``[('timestamp','<i8'),('x','<i2'),('y','<i2'),('polarity','i1')]``.

    Every option value of the reference is covered: ``additional_events_strategy`` 'slope' (the CLI's,
    v2ce.py:356), 'none', 'random' (LDATI.py:173-174: raw uniforms as offsets in seconds);
    ``pooling_type`` 'none' / 'avg' / 'weighted' with ``pooling_kernel_size`` (LDATI.py:177-182; only
    shapes 'slope'); ``bidirectional`` relocation (LDATI.py:107-122)."""
    assert pooling_type in ["avg", "weighted", "none"]                     # LDATI.py:135
    assert additional_events_strategy in ["none", "random", "slope"]       # LDATI.py:136
    ev = ldati_device(y, t0=t0, fps=fps, rng=rng, seed=seed, frame_base=frame_base, uniforms=uniforms,
                      strategy=additional_events_strategy, bidirectional=bidirectional,
                      pooling_type=pooling_type, pooling_kernel_size=pooling_kernel_size)
    if strict_reference_errors and ev.max_n == 0:
        raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0 "
                           "(reference LDATI.py:200 raises on an event-free chunk)")
    return ev.to_recarrays()
