"""CPU: v2ce.py glue arithmetic -- oracle/glue.py and the product glue against goldens captured
from the stub-imported reference v2ce.py (oracle/make_goldens.py gen_glue)."""
import os

import numpy as np
import pytest

from oracle import glue as OG


@pytest.fixture(scope="module")
def g7(gold_dir):
    return np.load(os.path.join(gold_dir, "glue_g7.npz"))


def test_sequence_plans(g7):
    for n in (17, 18, 33, 100, 2048):
        want = g7[f"plan_{n}"]
        num, mode, starts = OG.sequence_plan(n)
        assert [num, mode] + list(starts) == want.tolist()
    assert OG.sequence_plan(2048)[2][-1] == 2031


def test_frame_offsets(g7):
    for fps in (25, 30):
        want = g7[f"offsets_fps{fps}"]
        got = np.array([OG.frame_offset_us(i, fps) for i in range(4096)], np.int64)
        assert np.array_equal(got, want)
    # the double expression differs from exact floor(i*1e6/fps)
    assert OG.frame_offset_us(123, 30) == 4099999


def test_preprocess(g7):
    got = OG.preprocess(g7["frames"][:5])
    assert got.dtype == np.float32 and got.tobytes() == g7["pre5"].tobytes()


def test_pano_tiles():
    assert OG.pano_tiles(1384, 346) == [(0, 346, 0), (346, 692, 0), (692, 1038, 0), (1038, 1384, 0)]
    assert OG.pano_tiles(20, 12) == [(0, 12, 0), (8, 20, 8)]
    assert OG.center_crop_cols(20, 12) == (4, 16)


def test_merge_sources(g7):
    H, WF, width, N, bs = g7["params"].tolist()
    src = OG.merged_pair_sources(N)
    assert len(src) == N - 1 == g7["center"].shape[0]
    assert src[15] == (0, 15) and src[16] == (1, 13) and src[-1] == (1, 15)
