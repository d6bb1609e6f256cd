"""Host glue of the hot path, mirroring the functions of ``/root/reference/v2ce.py``.

Same names, argument meaning and results as the reference's ``image_pre_processing`` (v2ce.py:45),
``infer_center_image_unit`` (:66), ``infer_pano_image_unit`` (:91), ``video_to_voxels`` (:131) and
``merge_voxels`` (:211) -- except that nothing leaves the device between the model and LDATI: the
voxel grids stay in HBM as torch tensors (the reference round-trips them through host numpy,
v2ce.py:86,204,353).  Pinned by tests/golden/glue_g7.npz.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence

import numpy as np
import torch

logger = logging.getLogger("V2CE")

MEAN, STD = np.float32(0.153), np.float32(0.165)            # v2ce.py:54-55


def _linear_coeffs(n_in: int, n_out: int, zero_at_edges: bool):
    """OpenCV's INTER_LINEAR tap table for one axis (modules/imgproc/src/resize.cpp, cv::resize ->
    resizeGeneric_: ``scale = 1. / ((double)n_out / n_in)``; ``f = (float)((d + 0.5) * scale - 0.5)``;
    ``s = cvFloor(f)``; ``f -= s`` IN FLOAT).  Horizontal axis (`zero_at_edges`): ``s < 0 -> (f, s) = (0, 0)``,
    ``s >= n_in - 1 -> (f, s) = (0, n_in - 1)`` and the second tap is never read there (HResizeLinear copies
    ``S[s] * 1`` from xmax on).  Vertical axis: both rows are only CLIPPED to [0, n_in - 1], f is kept.
    Returns (first tap, second tap, weight of the second tap as f32)."""
    scale = 1.0 / (float(n_out) / float(n_in))                       # two f64 divisions, like OpenCV
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f.astype(np.float64)).astype(np.int64)               # cvFloor of the float value
    f = f - s.astype(np.float32)                                      # f32 subtraction
    if zero_at_edges:
        lo, hi = s < 0, s >= n_in - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, n_in - 1, s))
        return s, np.minimum(s + 1, n_in - 1), f
    return np.clip(s, 0, n_in - 1), np.clip(s + 1, 0, n_in - 1), f


def _resize_bilinear(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) for a float32 image with the default INTER_LINEAR, restated from
    OpenCV's published scalar algorithm (resize.cpp; the reference calls it at v2ce.py:57-58):
    identity when the size matches (every BASELINE config); an exact 2x2 decimation is OpenCV's INTER_AREA
    fast path (``(a + b + c + d) * 0.25``: cv::resize switches to it when both scales are exactly 2);
    otherwise taps / weights per axis from ``_linear_coeffs``, the HORIZONTAL pass ``S[s] * (1 - f) + S[s+1] * f``
    over the source rows first, then the vertical pass ``r0 * (1 - g) + r1 * g``, every operation a separately
    rounded f32 operation.  OpenCV itself is not installed here: pinned by hand-derived vectors
    (tests/test_product_glue.py), not against the library (whose SIMD builds may contract a*b+c)."""
    h, w = img.shape
    if (w, h) == (out_w, out_h):
        return img
    img = img.astype(np.float32, copy=False)
    if w == 2 * out_w and h == 2 * out_h:
        a, b, c, d = img[0::2, 0::2], img[0::2, 1::2], img[1::2, 0::2], img[1::2, 1::2]
        return (((a + b) + (c + d)) * np.float32(0.25)).astype(np.float32)
    x0, x1, fx = _linear_coeffs(w, out_w, True)
    y0, y1, fy = _linear_coeffs(h, out_h, False)
    one = np.float32(1)
    ax0, ax1 = (one - fx)[None, :], fx[None, :]
    rows = img[:, x0] * ax0 + img[:, x1] * ax1                        # [h, out_w]; at the right edge ax1 == 0
    rows[:, x0 == w - 1] = img[:, w - 1:w]                            # HResizeLinear's copy region (S[s] * 1)
    by0, by1 = (one - fy)[:, None], fy[:, None]
    return (rows[y0] * by0 + rows[y1] * by1).astype(np.float32)


def image_pre_processing(images: np.ndarray, height: int = 260) -> np.ndarray:
    """v2ce.py:45-64.  images [N,H,W] uint8 -> image units [N-1,2,height,W'] float32."""
    images = images.astype(np.float32) / 255
    images = np.stack([_resize_bilinear(img, int(img.shape[1] / img.shape[0] * height), height)
                       for img in images], axis=0)
    units = np.stack([images[:-1], images[1:]], axis=1)
    return ((units - MEAN) / STD).astype(np.float32)       # transforms.Normalize: sub then div, f32


def image_pre_processing_device(frames: torch.Tensor, height: Optional[int] = None) -> torch.Tensor:
    """Device twin of ``image_pre_processing``: frames [N,H,W] uint8 on the device -> [N-1,2,height,W']
    f32, W' = int(W/H*height) (v2ce.py:57), bit-identical to the host path (identity resize when the
    size already matches, bilinear otherwise)."""
    from . import hip
    if not frames.is_cuda or frames.dtype != torch.uint8 or frames.dim() != 3:
        raise hip.V2ceHipError("image_pre_processing_device: expected a uint8 [N,H,W] device tensor")
    frames = frames.contiguous()
    n, h, w = frames.shape
    oh = h if height is None else int(height)
    ow = int(w / h * oh)
    units = torch.empty((n - 1, 2, oh, ow), dtype=torch.float32, device=frames.device)
    with torch.cuda.device(frames.device):
        if (oh, ow) == (h, w):
            hip.check(hip.lib().v2ce_preprocess_pairs(frames.data_ptr(), n, h, w, float(MEAN), float(STD),
                                                      units.data_ptr(), hip.stream_ptr(frames.device)),
                      "v2ce_preprocess_pairs")
        else:
            hip.check(hip.lib().v2ce_preprocess_pairs_resize(frames.data_ptr(), n, h, w, oh, ow, float(MEAN), float(STD),
                                                             units.data_ptr(), hip.stream_ptr(frames.device)),
                      "v2ce_preprocess_pairs_resize")
    return units


def run_guarded(model, fn, comm=None):
    """fn() (a whole clip through ``model``) under the split-half range guard (include/v2ce_hip.h,
    VERDICT r1 #7): if a convolution reported a guard bound above the limit -- activations whose dynamic
    range one scale per tensor does not cover at the 1e-5 bar -- the spectral-norm state is rewound and
    fn() runs again on the exact-f32 kernels; the result is then that of precision='f32'.  The decision
    is the maximum over the ranks of ``comm`` (every rank must take the same branch: fn may contain
    collectives).  Models without the guard (test stand-ins) run fn() once."""
    if not hasattr(model, "range_guard_value") or getattr(model, "precision", None) != "f16x2":
        return fn()
    snap = model.sn_snapshot()
    model.range_guard_value()                              # clear what earlier calls left
    per_call, model.guard = getattr(model, "guard", "deferred"), "deferred"   # one check per clip, no per-call sync
    try:
        out = fn()
    finally:
        model.guard = per_call
    worst = model.range_guard_value()
    if comm is None:
        from . import dist as vdist
        comm = vdist.default_comm()
    worst = comm.max_float(worst, device=next(model.parameters()).device)
    if worst <= model.RANGE_GUARD_LIMIT:
        return out
    import logging
    logging.getLogger("V2CE").warning(
        f"split-half range guard: bound {worst:.3e} > {model.RANGE_GUARD_LIMIT:.1e}; repeating the clip on the exact-f32 kernels")
    del out
    model.sn_restore(snap)
    with model.exact_f32():                                # (precision 'f32': the guard mode is irrelevant)
        return fn()


def sequence_plan(frame_count: int, seq_len: int = 16):
    """v2ce.py:149-154 -> (sequence_num, mode, starting_indexes)."""
    if frame_count < seq_len + 1:
        raise ValueError(f"need at least {seq_len + 1} frames ({seq_len} frame-pairs); got {frame_count} "
                         "(the reference indexes frame -1 in that case, v2ce.py:150-154)")
    sequence_num = int(np.ceil((frame_count - 1) / seq_len))
    mode = (frame_count - 1) % seq_len
    starts = np.arange(sequence_num) * seq_len
    if mode != 0:
        starts[-1] -= (seq_len - mode)
    return sequence_num, mode, starts


def frame_offset_us(i: int, fps) -> int:
    """v2ce.py:365: ``int(i * 1 / fps * 1e6)`` -- double arithmetic left to right, truncated (this
    differs from floor(i*1e6/fps) for some i)."""
    return int(i * 1 / fps * 1e6)


@torch.no_grad()
def infer_center_image_unit(model, image_units: torch.Tensor, width: int = 346) -> torch.Tensor:
    """v2ce.py:66-89: centre crop on the width, one model call; result stays on the device."""
    fw = image_units.shape[-1]
    image_units = image_units[..., fw // 2 - width // 2: fw // 2 + width // 2]
    return model(image_units.float().contiguous())


@torch.no_grad()
def infer_pano_image_unit(model, image_units: torch.Tensor, width: int = 346) -> torch.Tensor:
    """v2ce.py:91-129: split the width into `width`-wide patches (last one = last `width` columns,
    keeping its trailing remainder), one model call per patch, concatenate on the width."""
    fw = image_units.shape[-1]
    patch_num = int(np.ceil(fw / width))
    exact_div = fw % 346 == 0            # sic: the reference hard-codes 346 here (v2ce.py:104)
    rem = fw % width
    outs = []
    for i in range(patch_num):
        last_partial = i == patch_num - 1 and not exact_div
        patch = image_units[..., -width:] if last_partial else image_units[..., i * width:(i + 1) * width]
        pred = model(patch.float().contiguous())
        if last_partial:
            pred = pred[..., -rem:]
        outs.append(pred)
    return torch.cat(outs, dim=-1)


def merge_voxels(voxel_list: Sequence[torch.Tensor], height: int, width: int, mode: int = 0) -> torch.Tensor:
    """v2ce.py:211-239 on device tensors: list of [b,16,20,H,W] batches -> [L,2,10,H,W]; of the
    overlapped last sequence only the last `mode` pairs are kept."""
    parts = [v.reshape(-1, 2, 10, height, width) for v in voxel_list[:-1]]
    last = voxel_list[-1]
    if last.shape[0] > 1:
        parts.append(last[:-1].reshape(-1, 2, 10, height, width))
    tail = last[-1][-mode:] if mode != 0 else last[-1]
    parts.append(tail.reshape(-1, 2, 10, height, width))
    return torch.cat(parts, dim=0)


@torch.no_grad()
def video_to_voxels(model, frames: Optional[np.ndarray] = None, read_frames=None, frame_count=None,
                    infer_type: str = "center", seq_len: int = 16, width: int = 346, height: int = 260,
                    batch_size: int = 1, device="cuda") -> torch.Tensor:
    """v2ce.py:131-209.  `frames` [N,H,W] uint8, or `read_frames(range)` + `frame_count` for a
    streaming source (re-readable: the range guard may run the clip twice, see run_guarded).  Returns the
    merged voxel grid [N-1,2,10,height,W'] on the device."""
    return run_guarded(model, lambda: _video_to_voxels(model, frames, read_frames, frame_count, infer_type,
                                                       seq_len, width, height, batch_size, device))


def _video_to_voxels(model, frames, read_frames, frame_count, infer_type, seq_len, width, height, batch_size, device):
    assert frames is not None or (read_frames is not None and frame_count is not None)
    if frames is not None:
        frame_count = len(frames)
        read_frames = lambda idx: frames[list(idx)]
    sequence_num, mode, starts = sequence_plan(frame_count, seq_len)
    logger.debug(f"Found {frame_count} images, divided into {sequence_num} sequences; mode {mode}")
    all_pred: List[torch.Tensor] = []
    pending = []
    out_width = width
    for seq_idx, start in enumerate(starts):
        images = np.asarray(read_frames(range(int(start), int(start) + seq_len + 1)))
        if images.dtype == np.uint8 and str(device).startswith("cuda"):
            # ship the u8 frames (4x fewer PCIe bytes); resize (v2ce.py:57) and normalise on the device
            units = image_pre_processing_device(torch.from_numpy(images).to(device, non_blocking=True), height)
            pending.append(units[None])
        else:
            units = image_pre_processing(images, height=height)
            pending.append(torch.from_numpy(units[np.newaxis]))
        if len(pending) == batch_size or seq_idx == len(starts) - 1:
            batch = torch.cat(pending, dim=0).to(device, non_blocking=True)
            if infer_type == "center":
                out_width = width
                pred = infer_center_image_unit(model, batch, width)
            elif infer_type == "pano":
                out_width = batch.shape[-1]
                pred = infer_pano_image_unit(model, batch, width)
            else:
                raise ValueError(f"Invalid infer_type {infer_type}")
            pending = []
            all_pred.append(pred)
    return merge_voxels(all_pred, height=height, width=out_width, mode=mode)
