"""ctypes front-end of the C LDATI oracle (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``sample_voxel_statistical_oracle`` mirrors the reference call
``scripts/LDATI.py:126 sample_voxel_statistical`` for every option value (``additional_events_strategy`` in 'slope' / 'none' / 'random', ``pooling_type`` in
'none' / 'avg' / 'weighted', ``bidirectional``) and returns the same list of packed
recarrays, but in the *stable* tie order (reference ``argsort`` is only stable for segments of
>= 32768 events; see ``canonicalize``).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EVENT_DTYPE = np.dtype([("timestamp", "<i8"), ("x", "<i2"), ("y", "<i2"), ("polarity", "i1")])
RNG_REPLAY, RNG_PHILOX = 0, 1


def build() -> str:
    """Compile oracle/libv2ce_oracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "libv2ce_oracle.so")
    src = os.path.join(_HERE, "ldati_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libv2ce_oracle.so"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i64p, f32p = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_float)
        L.v2ce_oracle_ldati_count.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64p,
                                              ctypes.POINTER(ctypes.c_int32)]
        L.v2ce_oracle_ldati_count.restype = ctypes.c_int
        L.v2ce_oracle_ldati_emit.argtypes = [
            f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
            ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, i64p, i64p,
            ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int16),
            ctypes.POINTER(ctypes.c_int8)]
        L.v2ce_oracle_ldati_emit.restype = ctypes.c_int
        L.v2ce_oracle_ldati_count2.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               i64p, ctypes.POINTER(ctypes.c_int32)]
        L.v2ce_oracle_ldati_count2.restype = ctypes.c_int
        L.v2ce_oracle_ldati_emit2.argtypes = [
            f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32p, ctypes.c_int, ctypes.c_uint64,
            ctypes.c_int64, i64p, i64p, ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int16),
            ctypes.POINTER(ctypes.c_int8)]
        L.v2ce_oracle_ldati_emit2.restype = ctypes.c_int
        L.v2ce_oracle_relocate2.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, i64p, f32p]
        L.v2ce_oracle_relocate2.restype = None
        L.v2ce_oracle_philox_fill.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_uint64, ctypes.c_int64]
        L.v2ce_oracle_philox_fill.restype = None
        L.v2ce_oracle_philox_uniform.argtypes = [ctypes.c_uint64] + [ctypes.c_uint32] * 4
        L.v2ce_oracle_philox_uniform.restype = ctypes.c_float
        L.v2ce_oracle_relocate.argtypes = [f32p, ctypes.c_int64, i64p, f32p]
        L.v2ce_oracle_relocate.restype = None
        _LIB = L
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def check_arange_len(fps: float) -> None:
    """The reference reshapes ``arange(0, 1/fps, 1/fps/9)`` to 9 bins (LDATI.py:163); any fps for
    which that arange does not have 9 elements raises inside the reference."""
    frame_step = 1 / fps
    voxel_step = 1 / fps / 9
    if math.ceil(frame_step / voxel_step) != 9:
        raise RuntimeError(f"arange(0, 1/{fps}, 1/{fps}/9) does not have 9 elements")


def relocate(y10: np.ndarray):
    """Per-pixel relocation of one 10-bin voxel column -> (counts[9] i64, debts[9] f32)."""
    y10 = np.ascontiguousarray(y10, dtype=np.float32)
    n = np.zeros(9, np.int64)
    d = np.zeros(9, np.float32)
    lib().v2ce_oracle_relocate(_p(y10, ctypes.c_float), 1, _p(n, ctypes.c_int64),
                               _p(d, ctypes.c_float))
    return n, d


STRATEGY = {"slope": 0, "none": 1, "random": 2}
POOLING = {"none": 0, "avg": 1, "weighted": 2}


def relocate2(y10: np.ndarray, bidirectional=False):
    """relocate with the bidirectional option (LDATI.py:107-122)."""
    y10 = np.ascontiguousarray(y10, dtype=np.float32)
    n = np.zeros(9, np.int64)
    d = np.zeros(9, np.float32)
    lib().v2ce_oracle_relocate2(_p(y10, ctypes.c_float), 1, int(bidirectional), _p(n, ctypes.c_int64),
                                _p(d, ctypes.c_float))
    return n, d


def count(vox: np.ndarray, strategy="slope", bidirectional=False):
    """vox [B,2,10,H,W] f32 -> (seg_counts [B,9] i64, max_n)."""
    vox = np.ascontiguousarray(vox, dtype=np.float32)
    B, P, C, H, W = vox.shape
    assert P == 2 and C == 10
    seg = np.zeros((B, 9), np.int64)
    mx = ctypes.c_int32(0)
    rc = lib().v2ce_oracle_ldati_count2(_p(vox, ctypes.c_float), B, H, W, STRATEGY[strategy], int(bidirectional),
                                        _p(seg, ctypes.c_int64), ctypes.byref(mx))
    assert rc == 0
    return seg, int(mx.value)


def philox_uniforms(B, H, W, max_n, seed, frame_base=0) -> np.ndarray:
    out = np.empty((B, 2, 9, H, W, max_n), np.float32)
    lib().v2ce_oracle_philox_fill(_p(out, ctypes.c_float), B, H, W, max_n, seed, frame_base)
    return out


def emit_soa(vox, fps=30, t0=0.0, uniforms=None, seed=0, frame_base=0, strategy="slope", bidirectional=False,
             pooling_type="none", pooling_kernel_size=3):
    """Run count + emit.  ``uniforms`` (dense [B,2,9,H,W,max_n] f32) selects REPLAY mode, else
    Philox with ``seed``.  Returns (seg_counts [B,9], ts, x, y, p)."""
    check_arange_len(fps)
    vox = np.ascontiguousarray(vox, dtype=np.float32)
    B, _, _, H, W = vox.shape
    seg, max_n = count(vox, strategy, bidirectional)
    offs = np.zeros(B * 9 + 1, np.int64)
    np.cumsum(seg.reshape(-1), out=offs[1:])
    total = int(offs[-1])
    ts = np.empty(total, np.int64)
    x = np.empty(total, np.int16)
    y = np.empty(total, np.int16)
    p = np.empty(total, np.int8)
    if uniforms is not None:
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert uniforms.shape[:5] == (B, 2, 9, H, W), uniforms.shape
        replay_max_n = uniforms.shape[5]
        assert replay_max_n >= max_n
        mode, uptr = RNG_REPLAY, _p(uniforms, ctypes.c_float)
    else:
        replay_max_n, mode, uptr = 0, RNG_PHILOX, None
    rc = lib().v2ce_oracle_ldati_emit2(_p(vox, ctypes.c_float), B, H, W, float(fps), float(t0), STRATEGY[strategy],
                                       int(bidirectional), POOLING[pooling_type], int(pooling_kernel_size), mode,
                                       uptr, replay_max_n, int(seed), int(frame_base),
                                       _p(offs, ctypes.c_int64), _p(ts, ctypes.c_int64),
                                       _p(x, ctypes.c_int16), _p(y, ctypes.c_int16),
                                       _p(p, ctypes.c_int8))
    if rc != 0:
        raise RuntimeError(f"v2ce_oracle_ldati_emit failed: {rc}")
    return seg, ts, x, y, p


def pack(ts, x, y, p) -> np.recarray:
    rec = np.empty(ts.shape[0], EVENT_DTYPE)
    rec["timestamp"], rec["x"], rec["y"], rec["polarity"] = ts, x, y, p
    return rec.view(np.recarray)


def sample_voxel_statistical_oracle(y, t0=0, fps=30, uniforms=None, seed=0, frame_base=0, strategy="slope",
                                    bidirectional=False, pooling_type="none", pooling_kernel_size=3):
    """Oracle twin of ``sample_voxel_statistical`` (LDATI.py:126): list[B] of packed recarrays."""
    vox = np.asarray(y, dtype=np.float32)
    seg, ts, x, yy, p = emit_soa(vox, fps=fps, t0=t0, uniforms=uniforms, seed=seed,
                                 frame_base=frame_base, strategy=strategy, bidirectional=bidirectional,
                                 pooling_type=pooling_type, pooling_kernel_size=pooling_kernel_size)
    per_frame = seg.sum(axis=1)
    ends = np.cumsum(per_frame)
    out = []
    for b in range(vox.shape[0]):
        lo, hi = int(ends[b] - per_frame[b]), int(ends[b])
        out.append(pack(ts[lo:hi], x[lo:hi], yy[lo:hi], p[lo:hi]))
    return out


def canonicalize(events: np.ndarray, seg_counts=None) -> np.ndarray:
    """Canonical order inside equal-timestamp runs of each (frame, bin) segment.

    The reference sorts each segment with ``argsort()`` (LDATI.py:297), which on CPU is the stable
    radix path only for >= 32768 elements and an (unstable) introsort below.  To compare an
    implementation that emits the stable order against reference output on small segments, both
    sides are re-sorted *within each segment* by (timestamp, polarity, y, x); events of one pixel
    and polarity with equal timestamps are indistinguishable records, so this is a total order on
    record values.  ``seg_counts`` (9 per frame) delimits the segments; if None the whole array is
    one segment.
    """
    ev = np.asarray(events)
    out = ev.copy()
    if seg_counts is None:
        seg_counts = [ev.shape[0]]
    lo = 0
    for n in np.asarray(seg_counts).reshape(-1):
        hi = lo + int(n)
        s = ev[lo:hi]
        order = np.lexsort((s["x"], s["y"], s["polarity"], s["timestamp"]))
        out[lo:hi] = s[order]
        lo = hi
    assert lo == ev.shape[0]
    return out
