"""The two ablation samplers of the reference's stage-2 study -- host side mirroring
``/root/reference/train/scripts/stage2/sample_methods/``:

* ``sample_voxel_baseline(y, t0, fps, even, random)``      random_even_sample.py:115-169
* ``sample_voxel_pure_slope(y, t0, fps, pooling_type, ...)``  pure_slope_sample.py:57-149 (named
  ``sample_voxel_statistical`` there; the LDATI sampler owns that name in this package)

Same signatures and return type (list of packed numpy recarrays in the order of
``np.sort(order='timestamp')``: lexicographic (timestamp, x, y, polarity)); computed by ``csrc/sampler.hip``
through the C ABI (``v2ce_sampler_count`` / ``v2ce_sampler_emit``).  No CPU path exists.

Differences from the reference, all deliberate:

* Random draws, as in ``LDATI.py`` of this package: ``rng='philox'`` (default; counter-based, independent of
  batching, keyed by ``seed``), ``rng='torch'`` (``torch.rand`` tensors of the reference's shapes drawn on the
  device and replayed, the Bernoulli event decided by ``u < frac(y)``), or caller-supplied ``u_int`` /
  ``u_dec`` / ``u_bern`` tensors (how parity with the oracle is tested).
* The pure-slope reference folds bin 9 into bin 8 in the CALLER's tensor (:92-93); here ``y`` is not modified.
* ``pooling_type`` 'avg' / 'weighted' of the pure-slope sampler (pure_slope_sample.py:79-85): the pooled values are
  sums of non-integer f32 numbers whose last bit depends on the convolution backend's summation order, so this one
  option is held to "the reference's events with timestamps within 1 us" instead of bit-exactness (measured on the
  goldens: identical; LDATI's pooled variant pools integer counts, which is exact, and is bit-exact in LDATI.py).
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional

import numpy as np
import torch

from . import hip
from .LDATI import DeviceEvents

C = 10


def _check_fps(fps) -> None:
    # random_even_sample.py:143 / pure_slope_sample.py:110 reshape arange(0, 1/fps, 1/fps/C) to C bins
    if math.ceil((1 / fps) / (1 / fps / C)) != C:
        raise RuntimeError(f"shape '[1, 1, {C}, 1, 1]' is invalid: arange(0, 1/{fps}, 1/{fps}/{C}) does not have "
                           f"{C} elements (the reference raises as well)")


def sampler_device(y: torch.Tensor, mode: int, t0=0, fps=30, *, rng: str = "philox", seed: Optional[int] = None,
                   frame_base: int = 0, u_int=None, u_dec=None, u_bern=None, pooling_type: str = "none",
                   pooling_kernel_size: int = 3) -> DeviceEvents:
    """count -> emit on the device; the events stay there (``DeviceEvents`` with one segment per frame)."""
    if y.dim() != 5 or y.shape[1] != 2 or y.shape[2] != C:
        raise ValueError(f"expected y of shape [B,2,{C},H,W], got {tuple(y.shape)}")
    _check_fps(fps)
    if not y.is_cuda:
        raise hip.V2ceHipError("sample_methods: y must be on a HIP device; no CPU path")
    if rng not in ("philox", "torch"):
        raise ValueError(f"rng must be 'philox' or 'torch', got {rng!r}")
    with torch.cuda.device(y.device):
        y = y.float().contiguous()
        B, _, _, H, W = y.shape
        dev, L, st = y.device, hip.lib(), hip.stream_ptr(y.device)
        opts = hip.SamplerOptions(mode=mode, rng_mode=hip.RNG_PHILOX, fps=float(fps), t0=float(t0), seed=0,
                                  frame_base=int(frame_base), replay_M=0, u_int=None, u_dec=None, u_bern=None)
        keep = []
        if mode == hip.SAMPLER_PURE_SLOPE and pooling_type != "none":     # pure_slope_sample.py:79-85: y_pooled (slope only)
            pooled = torch.empty_like(y)
            hip.check(L.v2ce_sampler_pool(y.data_ptr(), B, H, W, hip.POOL_WEIGHTED if pooling_type == "weighted" else hip.POOL_AVG,
                                          int(pooling_kernel_size), pooled.data_ptr(), st), "v2ce_sampler_pool")
            opts.pooled = pooled.data_ptr()
            keep.append(pooled)
        replay = u_bern is not None or rng == "torch"
        if replay:
            if u_bern is None:                      # the reference's own draws, on the device
                M = int(torch.floor(y).max().item()) if mode != hip.SAMPLER_PURE_SLOPE else \
                    int(torch.floor(torch.cat([y[:, :, :8], y[:, :, 8:9] + y[:, :, 9:10]], 2)).max().item())
                M = max(M, 0)
                if mode == hip.SAMPLER_RANDOM:      # random_even_sample.py:134, :149, then the bernoulli planes
                    u_int, u_dec = torch.rand([B, 2, C, H, W, M], device=dev), torch.rand([B, 2, C, H, W], device=dev)
                elif mode == hip.SAMPLER_PURE_SLOPE:   # pure_slope_sample.py:100 first, :121 after the bernoulli planes
                    u_dec = torch.rand([B, 2, C, H, W], device=dev)
                u_bern = torch.rand([B, 2, C, H, W], device=dev)
                if mode == hip.SAMPLER_PURE_SLOPE:
                    u_int = torch.rand([B, 2, C, H, W, M], device=dev)
            opts.rng_mode = hip.RNG_REPLAY
            u_bern = hip.require_device_f32(torch.as_tensor(u_bern).to(dev), "u_bern")
            assert tuple(u_bern.shape) == (B, 2, C, H, W), u_bern.shape
            opts.u_bern = u_bern.data_ptr()
            keep.append(u_bern)
            if mode != hip.SAMPLER_EVEN:
                if u_int is None or u_dec is None:
                    raise ValueError("replayed draws need u_int, u_dec and u_bern")
                u_int = hip.require_device_f32(torch.as_tensor(u_int).to(dev), "u_int")
                u_dec = hip.require_device_f32(torch.as_tensor(u_dec).to(dev), "u_dec")
                assert tuple(u_int.shape[:5]) == (B, 2, C, H, W) and tuple(u_dec.shape) == (B, 2, C, H, W)
                opts.replay_M = int(u_int.shape[5])
                opts.u_int = u_int.data_ptr() if u_int.numel() else None
                opts.u_dec = u_dec.data_ptr()
                keep += [u_int, u_dec]
        else:
            if seed is None:       # reproducible under torch.manual_seed, like the reference's draws
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            opts.seed = int(seed)
        meta = torch.empty(B + 1, dtype=torch.int64, device=dev)              # frame counts | max floor(y)
        hip.check(L.v2ce_sampler_count(y.data_ptr(), B, H, W, ctypes.byref(opts), meta.data_ptr(), meta[B:].data_ptr(), st),
                  "v2ce_sampler_count")
        host = meta.cpu().numpy()                                             # the one inherent synchronisation
        counts, max_int = host[:B].copy(), int(host[B:].view(np.int32)[0])
        if opts.rng_mode == hip.RNG_REPLAY and mode != hip.SAMPLER_EVEN and max_int > opts.replay_M:
            raise ValueError(f"u_int holds {opts.replay_M} draws per voxel, the largest floor(y) is {max_int}")
        total = int(counts.sum())
        if total >= 2 ** 32:
            raise hip.V2ceHipError(f"{total} events in one call: split the batch")
        ts = torch.empty(total, dtype=torch.int64, device=dev)
        x = torch.empty(total, dtype=torch.int16, device=dev)
        yy = torch.empty(total, dtype=torch.int16, device=dev)
        p = torch.empty(total, dtype=torch.int8, device=dev)
        ws = torch.empty(L.v2ce_sampler_workspace_bytes(total), dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        hip.check(L.v2ce_sampler_emit(y.data_ptr(), B, H, W, ctypes.byref(opts), total, hip.ptr(ts), hip.ptr(x), hip.ptr(yy),
                                      hip.ptr(p), ws.data_ptr(), ws.numel(), status.data_ptr(), st), "v2ce_sampler_emit")
        ev = DeviceEvents(None, counts.reshape(B, 1), max_int, soa=(ts, x, yy, p))
        ev._status = status
        ev._status_message = ("sample_methods: a timestamp left its frame's range (NaN / inf from a degenerate slope; "
                              "the reference's value is platform-defined there)")
        ws.record_stream(torch.cuda.current_stream(dev))
        for t in keep:
            t.record_stream(torch.cuda.current_stream(dev))
    return ev


def sample_voxel_baseline(y, t0=0, fps=30, even=False, random=False, *, rng: str = "philox", seed: Optional[int] = None,
                          frame_base: int = 0, u_int=None, u_dec=None, u_bern=None) -> List[np.recarray]:
    """Drop-in for random_even_sample.py:115: floor(y) events per voxel at uniform-random (``random``) or evenly
    spaced (``even``; wins when both are set, as in the reference) times inside the voxel's bin, plus one event with
    probability frac(y); all ten bins; per frame sorted by (timestamp, x, y, polarity)."""
    assert even or random                                                      # :116
    mode = hip.SAMPLER_EVEN if even else hip.SAMPLER_RANDOM
    return sampler_device(y, mode, t0, fps, rng=rng, seed=seed, frame_base=frame_base, u_int=u_int, u_dec=u_dec,
                          u_bern=u_bern).to_recarrays()


def sample_voxel_pure_slope(y, t0=0, fps=30, pooling_type="none", pooling_kernel_size=3,
                            additional_events_strategy="slope", *, rng: str = "philox", seed: Optional[int] = None,
                            frame_base: int = 0, u_int=None, u_dec=None, u_bern=None) -> List[np.recarray]:
    """Drop-in for pure_slope_sample.py:57 (``sample_voxel_statistical`` there): slope-distributed event times from
    the un-relocated voxel values, bin 9 folded into bin 8, Bernoulli rounding of the fractional part."""
    assert pooling_type in ["avg", "weighted", "none"]                         # :68
    assert additional_events_strategy in ["none", "random", "slope"]           # :69 (asserted, never used there)
    return sampler_device(y, hip.SAMPLER_PURE_SLOPE, t0, fps, rng=rng, seed=seed, frame_base=frame_base, u_int=u_int,
                          u_dec=u_dec, u_bern=u_bern, pooling_type=pooling_type,
                          pooling_kernel_size=pooling_kernel_size).to_recarrays()
