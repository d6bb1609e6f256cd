"""CPU restatement of the reference voxeliser (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``gen_discretized_event_volume`` follows
``/root/reference/train/scripts/utils/events_utils.py:118-175`` (``calc_floor_ceil_delta`` :118-126,
``create_update`` :128-145, ``gen_discretized_event_volume`` :147-175): events -> ``[2*bins, H, W]``
f32 volume, time axis rescaled to ``[0, bins-1]`` over the event set's own [t_min, t_max], every
event split linearly between its floor and ceil bin, positive polarity in the first ``bins`` planes and
negative (``polarity == 0``) in the second.  Arithmetic types as torch evaluates them on CPU:
``(bins-1) / (t_max-t_min)`` is ``reciprocal(t_max-t_min) * (bins-1)`` in f32 (``__rtruediv__``), the product with
``t - t_min`` (int64) is f32, ``+-1e-8`` are f32 additions, ``put_(accumulate=True)`` sums in f32
in event order (floor updates first, then ceil updates).  Pinned by tests/golden/voxelize_g8.npz
(generated from the reference by oracle/make_goldens.py).
"""
from __future__ import annotations

import numpy as np


def gen_discretized_event_volume(events: np.ndarray, vol_size) -> np.ndarray:
    nb2, H, W = (int(v) for v in vol_size)
    bins = nb2 // 2
    x = events["x"].astype(np.int64)
    y = events["y"].astype(np.int64)
    t = events["timestamp"].astype(np.int64)
    neg = events["polarity"] == 0                                   # :155 p[p == 0] = -1
    t_min, t_max = t.min(), t.max()
    # :159 `python_int / int64_tensor` is Tensor.__rtruediv__ = reciprocal(tensor) * int, in f32 [probed]
    scale = np.float32(np.float32(1) / np.float32(t_max - t_min)) * np.float32(bins - 1)
    ts = (t - t_min).astype(np.float32) * scale                     # int64 * f32 -> f32
    ts = np.clip(ts, np.float32(0), np.float32(bins - 1))           # :160
    fl = np.floor(ts + np.float32(1e-8))                            # :119
    ce = np.ceil(ts - np.float32(1e-8))                             # :120
    ce_fake = np.floor(ts) + np.float32(1)                          # :121
    d_ce = (ts - fl).astype(np.float32)                             # :123
    d_fl = (ce_fake - ts).astype(np.float32)                        # :124
    assert (x >= 0).all() and (x < W).all() and (y >= 0).all() and (y < H).all()      # :129-130
    assert (fl >= 0).all() and (ce < bins).all()                                      # :131
    plane = np.where(neg, bins, 0).astype(np.int64)                 # :134-136
    vol = np.zeros(nb2 * H * W, np.float32)
    for tb, val in ((fl.astype(np.int64), d_fl), (ce.astype(np.int64), d_ce)):         # :164-173
        idx = (H * W) * (tb + plane) + W * y + x                    # :138-140
        np.add.at(vol, idx, val)        # sequential f32 accumulation in event order, like put_
    return vol.reshape(nb2, H, W)
