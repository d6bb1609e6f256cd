"""Events -> discretised event volume on the device (the inverse of LDATI).

Drop-in for ``gen_discretized_event_volume(events, vol_size)`` of
``train/scripts/utils/events_utils.py:147-175`` (the authors' round-trip sanity check,
``train/scripts/stage2/stage2_metrics.py:187-190``): same argument meaning, the volume comes back as
a float32 torch tensor ``[2*bins, H, W]`` -- on the device here.  There is no CPU path.
"""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np
import torch

from . import hip
from .LDATI import EVENT_DTYPE, DeviceEvents


def _soa(events, device):
    if isinstance(events, DeviceEvents):
        return events.ts, events.x, events.y, events.p
    if isinstance(events, (tuple, list)) and len(events) == 4 and all(torch.is_tensor(e) for e in events):
        return tuple(events)
    ev = np.asarray(events)
    if ev.dtype.names is None or not {"timestamp", "x", "y", "polarity"} <= set(ev.dtype.names):
        raise TypeError("events must be a structured array with fields timestamp, x, y, polarity "
                        "(the LDATI record dtype), a DeviceEvents or a (ts, x, y, p) tuple of device tensors")
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt, copy=False))).to(device)
    return (to(ev["timestamp"], np.int64), to(ev["x"], np.int16), to(ev["y"], np.int16), to(ev["polarity"], np.int8))


def gen_discretized_event_volume(events: Union[np.ndarray, DeviceEvents, Sequence[torch.Tensor]], vol_size,
                                 device="cuda") -> torch.Tensor:
    """events_utils.py:147-175.  ``vol_size = (2*bins, H, W)``."""
    nb2, H, W = (int(v) for v in vol_size)
    if nb2 < 4 or nb2 % 2:
        raise ValueError(f"vol_size[0] must be an even number >= 4, got {nb2}")
    if not isinstance(events, (DeviceEvents, tuple, list)) and len(events) == 0:
        raise RuntimeError("gen_discretized_event_volume: no events (the reference raises on t.min() of an empty tensor)")
    ts, x, y, p = _soa(events, device)
    for t, dt, name in ((ts, torch.int64, "timestamp"), (x, torch.int16, "x"), (y, torch.int16, "y"), (p, torch.int8, "polarity")):
        if not t.is_cuda:
            raise hip.V2ceHipError(f"{name} must live on a HIP device; there is no CPU path")
        if t.dtype != dt or not t.is_contiguous():
            raise TypeError(f"{name} must be a contiguous {dt} tensor")
    n = int(ts.shape[0])
    if n == 0:
        raise RuntimeError("gen_discretized_event_volume: no events (the reference raises on t.min() of an empty tensor)")
    # events_utils.py:129-130 assert the coordinates: checked on the device, fetched together with the
    # time range in ONE small D2H copy after the launch
    bad = ((x < 0) | (x >= W) | (y < 0) | (y >= H)).any().to(torch.int64).reshape(1)
    vol = torch.empty((nb2, H, W), dtype=torch.float32, device=ts.device)
    rng = torch.empty(2, dtype=torch.int64, device=ts.device)
    with torch.cuda.device(ts.device):
        hip.check(hip.lib().v2ce_voxelize_events(ts.data_ptr(), x.data_ptr(), y.data_ptr(), p.data_ptr(), n, nb2 // 2,
                                                 H, W, vol.data_ptr(), rng.data_ptr(), hip.stream_ptr(ts.device)),
                  "v2ce_voxelize_events")
    t_min, t_max, is_bad = (int(v) for v in torch.cat([rng, bad]).tolist())
    if is_bad:
        raise AssertionError("gen_discretized_event_volume: event coordinates outside the volume")
    if t_max == t_min:
        raise RuntimeError("gen_discretized_event_volume: t_max == t_min (the reference divides by zero here)")
    return vol
