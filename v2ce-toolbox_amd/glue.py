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


def _resize_bilinear(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) with the default INTER_LINEAR convention (half-pixel centres,
    edge clamp) for float32 images.  Identity when the size already matches (the 260-high inputs
    of every BASELINE config); cv2 is not installed here, so the non-identity case is unpinned."""
    h, w = img.shape
    if (w, h) == (out_w, out_h):
        return img
    ys = (np.arange(out_h, dtype=np.float64) + 0.5) * (h / out_h) - 0.5
    xs = (np.arange(out_w, dtype=np.float64) + 0.5) * (w / out_w) - 0.5
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0).astype(np.float32)[:, None]
    fx = (xs - x0).astype(np.float32)[None, :]
    y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
    x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
    top = img[y0c][:, x0c] * (1 - fx) + img[y0c][:, x1c] * fx
    bot = img[y1c][:, x0c] * (1 - fx) + img[y1c][:, x1c] * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


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
    out = fn()
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
    with model.exact_f32():
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
