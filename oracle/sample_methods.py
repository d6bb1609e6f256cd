"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, one separately rounded f32 operation per
reference operation) of the two ablation samplers of SURVEY 8f4:

* ``sample_voxel_baseline``       /root/reference/train/scripts/stage2/sample_methods/random_even_sample.py:115-169
* ``sample_voxel_pure_slope``     /root/reference/train/scripts/stage2/sample_methods/pure_slope_sample.py:57-149
  (the reference calls it ``sample_voxel_statistical`` too; renamed here because the LDATI sampler owns that name)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

Pinned by tests/golden/sampler_g10_*.npz: the reference itself run in the build container
(oracle/make_goldens.py gen_sample_methods) with ``torch.rand`` recorded and ``torch.bernoulli(p)``
routed through a recorded uniform draw (``u < p``, which is what the CPU kernel of torch computes
per element, from its own generator), so that the random draws are data of the fixture.

Random draws are explicit arguments, shaped like the reference's tensors:
  u_int  [B,2,10,H,W,M]  timestamps of the floor(y) events   (M = int(floor(y).max()))
  u_dec  [B,2,10,H,W]    timestamp of the Bernoulli(frac(y)) event
  u_bern [B,2,10,H,W]    the Bernoulli draw: the event exists iff u_bern < frac(y)
Output order: ``np.sort(records, order='timestamp')`` compares the remaining fields in dtype order after
the timestamp, so the result is the lexicographic order of (timestamp, x, y, polarity) -- canonical.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import ldati as O

EVENT_DTYPE = O.EVENT_DTYPE
F = np.float32
C = 10
KIND_INT, KIND_DEC, KIND_BERN = 0, 1, 2


def offsets(fps: float, t0: float) -> np.ndarray:
    """``torch.arange(0, 1/fps, 1/fps/C) + t0`` (random_even_sample.py:143, pure_slope_sample.py:110): arange
    evaluates start + i*step in double and rounds to f32; ``+ t0`` is an f32 addition."""
    step = 1 / fps / C
    n = int(np.ceil((1 / fps) / step))
    if n != C:
        raise RuntimeError(f"shape '[1, 1, {C}, 1, 1]' is invalid for input of size {n}")
    return (np.arange(C, dtype=np.float64) * step).astype(F) + F(t0)


def _to_us(ts: np.ndarray, off: np.ndarray, axis: int) -> np.ndarray:
    shape = [1] * ts.ndim
    shape[axis] = C
    ts = ts + off.reshape(shape)                 # ts += arange + t0
    ts = ts * F(1e6)                             # ts *= 1e6
    with np.errstate(invalid="ignore"):
        return ts.astype(np.int64)               # .to(torch.long): truncation


def _collect(ts_int, n_int, ts_dec, sel_dec):
    """pick_and_sort (both forms) + concatenate + np.sort(order='timestamp') for every frame."""
    B, P, _, H, W = n_int.shape
    yy, xx = np.meshgrid(np.arange(H, dtype=np.int16), np.arange(W, dtype=np.int16), indexing="ij")
    out = []
    for b in range(B):
        recs = []
        for pi in range(P):
            pol = 1 - pi                          # P index 0 = positive (polarity 1), index 1 = negative (0)
            for c in range(C):
                if ts_int.shape[-1]:
                    sel = np.arange(ts_int.shape[-1])[None, None, :] < n_int[b, pi, c][:, :, None]
                    hh, ww, _ = np.nonzero(sel)
                    recs.append((ts_int[b, pi, c][sel], xx[hh, ww], yy[hh, ww], pol))
                sd = sel_dec[b, pi, c]
                recs.append((ts_dec[b, pi, c][sd], xx[sd], yy[sd], pol))
        n = sum(len(r[0]) for r in recs)
        ev = np.empty(n, EVENT_DTYPE)
        lo = 0
        for t, x, y, pol in recs:
            ev["timestamp"][lo:lo + len(t)] = t
            ev["x"][lo:lo + len(t)] = x
            ev["y"][lo:lo + len(t)] = y
            ev["polarity"][lo:lo + len(t)] = pol
            lo += len(t)
        order = np.lexsort((ev["polarity"], ev["y"], ev["x"], ev["timestamp"]))
        out.append(ev[order].view(np.recarray))
    return out


def sample_voxel_baseline(y, t0=0, fps=30, even=False, random=False, u_int=None, u_dec=None, u_bern=None):
    """random_even_sample.py:115-169.  ``even`` wins when both flags are set (its assignment comes second)."""
    assert even or random                                                   # :116
    y = np.asarray(y, dtype=F)
    B, P, Cc, H, W = y.shape
    assert Cc == C
    delta = F(1 / (fps * C))                                                # :121, used as an f32 scalar
    ip = np.floor(y)                                                        # :125
    dp = y - ip                                                             # :126
    M = int(ip.max()) if ip.size else 0                                     # :129
    off = offsets(fps, t0)
    if even:
        j = np.arange(M, dtype=F)
        ts = (j / (ip[..., None] + F(1))) * delta                           # :138-140
    else:
        ts = np.asarray(u_int, F).reshape(B, P, C, H, W, -1)[..., :M] * delta   # :134
    ts_int = _to_us(ts, off, 2)                                             # :143-145
    if even:
        td = (ip / (ip + F(1))) * delta                                     # :152-153
    else:
        td = np.asarray(u_dec, F).reshape(B, P, C, H, W) * delta            # :149
    ts_dec = _to_us(td, off, 2)                                             # :156-158
    sel = np.asarray(u_bern, F).reshape(B, P, C, H, W) < dp                 # :53 (bernoulli)
    return _collect(ts_int, ip, ts_dec, sel)


def _slope_kb(y, fps):
    """pure_slope_sample.py:13-55 and :88-91 with pooling 'none', every step in f32."""
    vs = 1 / (fps * C)
    pad = np.concatenate([y[:, :, 1:2], y, y[:, :, C - 2:C - 1]], axis=2)   # reflect pad of the bin axis (:24)
    sum_xy = pad[:, :, 2:] - pad[:, :, :-2]                                 # conv1d [-1, 0, 1] (:38): one rounding
    k0 = (F(3) * sum_xy) / F(6)                                             # :52 with sum_x = 0, sum_x2 = 2, N = 3
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        k = (k0 / F(vs ** 2)) / (y + F(1e-8))                               # :88
        bb = F(1 / vs) - (F(vs) * k) / F(2)                                 # :91
    return k, bb


def _slope_ts(k, bb, u, fps):
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        t = (-bb + np.sqrt(bb * bb + (F(2) * k) * u)) / k                   # :105 / :130
        return np.where(k == 0, (u / F(fps)) / F(C), t)                     # :106 / :131


def pool_voxels(y, pooling_type, pooling_kernel_size=3):
    """pure_slope_sample.py:79-85: the voxel values pooled over the k x k pixel neighbourhood, zero padding.
    'weighted': F.conv2d with [[1,2,1],[2,4,2],[1,2,1]]/16 (:80-82); 'avg': nn.AvgPool2d(k, stride 1, padding k//2),
    i.e. window sum / k^2 (:84, count_include_pad).  Unlike LDATI's pooling (integer counts: exact) these are sums of
    arbitrary f32 values, whose last bit depends on the backend's summation order; here: taps in row-major order,
    one f32 multiply and one f32 add per tap.  Parity bar for everything derived from it: timestamps within 1 us
    (tests/test_oracle_samplers.py::test_pooled_pure_slope_close_to_reference)."""
    y = np.asarray(y, F)
    if pooling_type == "none":
        return y
    H, W = y.shape[-2:]
    if pooling_type == "weighted":
        r, wt = 1, (np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], F) / F(16))
    else:
        r = pooling_kernel_size // 2
        wt = np.ones((pooling_kernel_size, pooling_kernel_size), F)
    pad = np.zeros(y.shape[:-2] + (H + 2 * r, W + 2 * r), F)
    pad[..., r:r + H, r:r + W] = y
    acc = np.zeros_like(y)
    for dh in range(2 * r + 1):
        for dw in range(2 * r + 1):
            tap = pad[..., dh:dh + H, dw:dw + W]
            acc = acc + (wt[dh, dw] * tap if pooling_type == "weighted" else tap)
    return acc if pooling_type == "weighted" else acc / F(pooling_kernel_size * pooling_kernel_size)


def sample_voxel_pure_slope(y, t0=0, fps=30, pooling_type="none", pooling_kernel_size=3,
                            additional_events_strategy="slope", u_int=None, u_dec=None, u_bern=None):
    """pure_slope_sample.py:57-149.  Does not modify ``y`` (the reference folds bin 9 into bin 8 in place, :92-93,
    which reaches the caller's tensor).  Pooling shapes only the slope parameters (:79-91, from the UNFOLDED values);
    the event counts come from ``y`` itself."""
    assert pooling_type in ["avg", "weighted", "none"]
    assert additional_events_strategy in ["none", "random", "slope"]
    y = np.array(y, dtype=F)
    B, P, Cc, H, W = y.shape
    assert Cc == C
    k, bb = _slope_kb(pool_voxels(y, pooling_type, pooling_kernel_size), fps)
    y[:, :, 8] = y[:, :, 8] + y[:, :, 9]                                    # :92
    y[:, :, 9] = 0                                                          # :93
    ip = np.floor(y).astype(np.int32)                                       # :95
    dp = y - ip.astype(F)                                                   # :96
    off = offsets(fps, t0)
    ts_dec = _to_us(_slope_ts(k, bb, np.asarray(u_dec, F).reshape(B, P, C, H, W), fps), off, 2)   # :100-112
    sel = np.asarray(u_bern, F).reshape(B, P, C, H, W) < dp
    M = int(ip.max()) if ip.size else 0                                     # :117
    u = np.asarray(u_int, F).reshape(B, P, C, H, W, -1)[..., :M]
    ts_int = _to_us(_slope_ts(k[..., None], bb[..., None], u, fps), off, 2)  # :121-140
    return _collect(ts_int, ip, ts_dec, sel)


def philox_draws(B, H, W, M, seed, frame_base=0):
    """The uniforms the device draws in Philox mode, materialised densely: counter (pixel, j >> 2,
    32*kind + 10*P + c, frame_base + b), key = seed (csrc/sampler.hip)."""
    fn = O.lib().v2ce_oracle_philox_uniform
    fn.restype = ctypes.c_float
    fn.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    u_int = np.empty((B, 2, C, H, W, M), F)
    u_dec = np.empty((B, 2, C, H, W), F)
    u_bern = np.empty((B, 2, C, H, W), F)
    for b in range(B):
        for pi in range(2):
            for c in range(C):
                pc = 10 * pi + c
                for px in range(H * W):
                    h, w = divmod(px, W)
                    for j in range(M):
                        u_int[b, pi, c, h, w, j] = fn(seed, px, j, 32 * KIND_INT + pc, frame_base + b)
                    u_dec[b, pi, c, h, w] = fn(seed, px, 0, 32 * KIND_DEC + pc, frame_base + b)
                    u_bern[b, pi, c, h, w] = fn(seed, px, 0, 32 * KIND_BERN + pc, frame_base + b)
    return u_int, u_dec, u_bern


def events_close(a, b, tol_us=1):
    """Two event lists of one frame that may differ by a rounding step of a slope parameter: the same (x, y,
    polarity) multiset and, column by column in time order, timestamps within `tol_us`.  Returns the number of
    timestamps that differ (or -1 if the lists are not comparable)."""
    a, b = np.asarray(a), np.asarray(b)
    if len(a) != len(b):
        return -1
    def canon(e):
        o = np.lexsort((e["timestamp"], e["polarity"], e["y"], e["x"]))
        return e[o]
    a, b = canon(a), canon(b)
    if not (np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]) and np.array_equal(a["polarity"], b["polarity"])):
        return -1
    d = np.abs(a["timestamp"].astype(np.int64) - b["timestamp"].astype(np.int64))
    return -1 if d.size and d.max() > tol_us else int((d != 0).sum())
