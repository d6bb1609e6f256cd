"""LDATI (stage 2) -- host side mirroring ``/root/reference/scripts/LDATI.py``.

``sample_voxel_statistical`` keeps the reference signature (``scripts/LDATI.py:126``) and return
type (list of packed numpy recarrays, ``scripts/LDATI.py:308``); everything is computed by the HIP
kernels of ``csrc/ldati.hip`` through the C ABI (``include/v2ce_hip.h``).  No CPU path exists.

Differences from the reference, all deliberate (DESIGN.md "Stage 2"):

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

import math
from typing import List, Optional

import numpy as np
import torch

from . import hip

EVENT_DTYPE = np.dtype([("timestamp", "<i8"), ("x", "<i2"), ("y", "<i2"), ("polarity", "i1")])
assert EVENT_DTYPE.itemsize == 13


class DeviceEvents:
    """Events of one LDATI call, resident on the device (SoA) + host-side segment table."""

    def __init__(self, ts, x, y, p, seg_counts, max_n):
        self.ts, self.x, self.y, self.p = ts, x, y, p
        self.seg_counts = seg_counts            # np.int64 [B, 9]
        self.max_n = max_n

    @property
    def num_events(self) -> int:
        return int(self.ts.shape[0])

    @property
    def frame_counts(self) -> np.ndarray:
        return self.seg_counts.sum(axis=1)

    def packed(self) -> torch.Tensor:
        """uint8 [N*13] device tensor of packed records (LDATI.py:308 dtype)."""
        n = self.num_events
        out = torch.empty(max(n * 13, 4), dtype=torch.uint8, device=self.ts.device)
        if n:
            L = hip.lib()
            with torch.cuda.device(self.ts.device):
                hip.check(L.v2ce_events_pack(hip.ptr(self.ts), hip.ptr(self.x), hip.ptr(self.y),
                                             hip.ptr(self.p), n, hip.ptr(out),
                                             hip.stream_ptr(self.ts.device)), "v2ce_events_pack")
        return out[: n * 13]

    def to_recarrays(self) -> List[np.recarray]:
        ev = np.ascontiguousarray(self.packed().cpu().numpy()).view(EVENT_DTYPE)
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


def ldati_device(y: torch.Tensor, t0=0, fps=30, *, rng: str = "philox", seed: Optional[int] = None,
                 frame_base: int = 0, uniforms: Optional[torch.Tensor] = None,
                 frame_ts_add: Optional[torch.Tensor] = None, profile: Optional[list] = None,
                 path: str = "bucket", strategy: str = "slope") -> DeviceEvents:
    """Run count -> scan -> emit on the device and leave the events there.

    y: [B,2,10,H,W] on a HIP device.  One host synchronisation (reading the B*9+1 segment offsets
    and max_n) is inherent to the variable-length output.
    """
    if y.dim() != 5 or y.shape[1] != 2 or y.shape[2] != 10:
        raise ValueError(f"expected y of shape [B,2,10,H,W], got {tuple(y.shape)}")
    _check_fps(fps)
    if not y.is_cuda:
        raise hip.V2ceHipError("sample_voxel_statistical: y must be on a HIP device; no CPU path")
    # the C ABI launches on the current device's stream: make y's device current for the call
    with torch.cuda.device(y.device):
        return _ldati_device(y, t0, fps, rng, seed, frame_base, uniforms, frame_ts_add, profile, path, strategy)


def _ldati_device(y, t0, fps, rng, seed, frame_base, uniforms, frame_ts_add, profile, path, strategy):
    y = y.float().contiguous()                         # LDATI.py:143 `.float()`
    B, _, _, H, W = y.shape
    dev = y.device
    L = hip.lib()
    st = hip.stream_ptr(dev)
    if L.v2ce_ldati_lds_bytes(float(fps), float(t0)) == 0:
        raise hip.V2ceHipError(f"fps={fps}, t0={t0}: time bin too wide for the LDS key histogram")
    seg_counts = torch.empty(B * 9, dtype=torch.int64, device=dev)
    max_n_t = torch.empty(1, dtype=torch.int32, device=dev)
    offsets = torch.empty(B * 9 + 2, dtype=torch.int64, device=dev)
    strat = {"slope": hip.STRATEGY_SLOPE, "none": hip.STRATEGY_NONE}[strategy]
    hip.check(L.v2ce_ldati_count(y.data_ptr(), B, H, W, strat, seg_counts.data_ptr(), max_n_t.data_ptr(), st),
              "v2ce_ldati_count")
    hip.check(L.v2ce_ldati_scan(seg_counts.data_ptr(), B, offsets.data_ptr(), st), "v2ce_ldati_scan")
    offsets[B * 9 + 1:] = max_n_t.to(torch.int64)
    host = offsets.cpu().numpy()                       # the one sync
    offs, max_n = host[:B * 9 + 1], int(host[B * 9 + 1])
    total = int(offs[-1])
    segc = np.diff(offs).reshape(B, 9)

    mode, u_ptr, replay_max_n = hip.RNG_PHILOX, None, 0
    keep = None
    if uniforms is not None:
        uniforms = hip.require_device_f32(uniforms.to(dev), "uniforms")
        if tuple(uniforms.shape[:5]) != (B, 2, 9, H, W) or uniforms.shape[5] < max_n:
            raise ValueError(f"uniforms must be [B,2,9,H,W,>=max_n={max_n}], got {tuple(uniforms.shape)}")
        mode, u_ptr, replay_max_n, keep = hip.RNG_REPLAY, uniforms.data_ptr(), uniforms.shape[5], uniforms
    elif rng == "torch":
        keep = torch.rand([B, 2, 9, H, W, max_n], device=dev)      # LDATI.py:171
        mode, u_ptr, replay_max_n = hip.RNG_REPLAY, keep.data_ptr(), max_n
    elif rng == "philox":
        if seed is None:   # reproducible under torch.manual_seed, like the reference's draw
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    else:
        raise ValueError(f"rng must be 'philox' or 'torch', got {rng!r}")

    ts = torch.empty(total, dtype=torch.int64, device=dev)
    x = torch.empty(total, dtype=torch.int16, device=dev)
    yy = torch.empty(total, dtype=torch.int16, device=dev)
    p = torch.empty(total, dtype=torch.int8, device=dev)
    if total:
        add_ptr = None
        if frame_ts_add is not None:
            frame_ts_add = frame_ts_add.to(device=dev, dtype=torch.int64).contiguous()
            assert frame_ts_add.numel() == B
            add_ptr = frame_ts_add.data_ptr()
        max_seg = int(segc.max())
        ws, ws_bytes = None, 0
        if path == "bucket":       # pixel-parallel bucketed path; "sweep" = one workgroup per segment
            ws_bytes = L.v2ce_ldati_workspace_bytes(B, H, W, float(fps), float(t0), total, max_seg)
            if ws_bytes:
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        elif path != "sweep":
            raise ValueError(f"path must be 'bucket' or 'sweep', got {path!r}")
        if profile is not None:    # HIP events on the launch stream around the emit kernels
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        hip.check(L.v2ce_ldati_emit(y.data_ptr(), B, H, W, float(fps), float(t0), strat, mode, u_ptr,
                                    int(replay_max_n), int(seed or 0) & (2 ** 64 - 1), int(frame_base),
                                    offsets.data_ptr(), add_ptr, ts.data_ptr(), x.data_ptr(),
                                    yy.data_ptr(), p.data_ptr(), total, max_seg, hip.ptr(ws),
                                    int(ws_bytes), st), "v2ce_ldati_emit")
        if profile is not None:
            e1.record()
            # algorithmic bytes: 80 B per pixel read once + 13 B per event written once
            profile.append(("emit", e0, e1, 80 * B * H * W + 13 * total))
    ev = DeviceEvents(ts, x, yy, p, segc, max_n)
    ev._keepalive = (y, keep, frame_ts_add, offsets, ws if total else None)
    return ev


def sample_voxel_statistical(y, t0=0, fps=30, pooling_type="none", pooling_kernel_size=3,
                             additional_events_strategy="slope", bidirectional=False, *,
                             rng: str = "philox", seed: Optional[int] = None, frame_base: int = 0,
                             uniforms=None, strict_reference_errors: bool = False):
    """Drop-in for ``scripts/LDATI.py:126``: y [B,2,10,H,W] -> list[B] of packed recarrays
    ``[('timestamp','<i8'),('x','<i2'),('y','<i2'),('polarity','i1')]``.

    Supports the option values the reference CLI uses (v2ce.py:356); the other values of the
    reference's option strings raise NotImplementedError.
    """
    assert pooling_type in ["avg", "weighted", "none"]                     # LDATI.py:135
    assert additional_events_strategy in ["none", "random", "slope"]       # LDATI.py:136
    if pooling_type != "none" or additional_events_strategy == "random" or bidirectional:
        raise NotImplementedError(
            "the HIP path implements pooling_type='none', additional_events_strategy in "
            "{'slope' (v2ce.py:356), 'none'}, bidirectional=False")
    ev = ldati_device(y, t0=t0, fps=fps, rng=rng, seed=seed, frame_base=frame_base, uniforms=uniforms,
                      strategy=additional_events_strategy)
    if strict_reference_errors and ev.max_n == 0:
        raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0 "
                           "(reference LDATI.py:200 raises on an event-free chunk)")
    return ev.to_recarrays()
