"""Oracle restatement of the ``v2ce.py`` glue arithmetic (TEST INFRASTRUCTURE; numpy only).

Each function cites the reference lines it follows; pinned by tests/golden/glue_*.npz, which were
captured from the stub-imported reference ``v2ce.py`` (oracle/make_goldens.py).
"""
from __future__ import annotations

import numpy as np


def sequence_plan(frame_count: int, seq_len: int = 16):
    """v2ce.py:149-154 -> (sequence_num, mode, starting_indexes)."""
    sequence_num = int(np.ceil((frame_count - 1) / seq_len))
    mode = (frame_count - 1) % seq_len
    starts = np.arange(sequence_num) * seq_len
    if mode != 0:
        starts[-1] -= (seq_len - mode)
    return sequence_num, mode, starts


def preprocess(images_u8: np.ndarray) -> np.ndarray:
    """v2ce.py:45-64 for frames already at the target height (cv2.resize is then the identity):
    u8 -> f32/255, pair stacking [N-1,2,H,W], Normalize(0.153, 0.165) in f32."""
    img = images_u8.astype(np.float32) / 255
    units = np.stack([img[:-1], img[1:]], axis=1)
    return (units - np.float32(0.153)) / np.float32(0.165)


def center_crop_cols(full_width: int, width: int = 346):
    """v2ce.py:78 -> (lo, hi) column slice."""
    return full_width // 2 - width // 2, full_width // 2 + width // 2


def pano_tiles(full_width: int, width: int = 346):
    """v2ce.py:103-111,121-122 -> list of (in_lo, in_hi, keep_last) per tile; keep_last = number of
    trailing output columns kept (0 = all).  Quirk kept: exact_div tests ``% 346`` (v2ce.py:104)."""
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


def merged_pair_sources(frame_count: int, seq_len: int = 16):
    """v2ce.py:211-239 merge_voxels as an index map: for each output frame-pair, the
    (sequence index, pair index inside the sequence) it is taken from."""
    sequence_num, mode, _ = sequence_plan(frame_count, seq_len)
    src = []
    for s in range(sequence_num - 1):
        src += [(s, j) for j in range(seq_len)]
    last = sequence_num - 1
    if mode != 0:
        src += [(last, j) for j in range(seq_len - mode, seq_len)]
    else:
        src += [(last, j) for j in range(seq_len)]
    return src


def frame_offset_us(i: int, fps) -> int:
    """v2ce.py:365 ``int(i * 1 / fps * 1e6)`` (double arithmetic, left to right, truncated)."""
    return int(i * 1 / fps * 1e6)
