"""CPU: host glue of the product (v2ce-toolbox_amd/glue.py, dist.py) against the goldens captured
from the reference v2ce.py and against the oracle restatement, with a stand-in model."""
import os

import numpy as np
import pytest
import torch

from oracle import glue as OG
from v2ce_toolbox_amd import dist as vdist
from v2ce_toolbox_amd import glue


@pytest.fixture(scope="module")
def g7(gold_dir):
    return np.load(os.path.join(gold_dir, "glue_g7.npz"))


class FakeModel:
    """Deterministic stand-in with the call-index dependence of the real model (spectral norm)."""

    def __init__(self):
        self.calls = 0

    def advance_spectral_norm(self):
        self.calls += 1

    def __call__(self, x):                        # [B,L,2,H,W] -> [B,L,20,H,W]
        B, L, _, H, W = x.shape
        base = x.mean(dim=2, keepdim=True) + 0.01 * self.calls
        ch = torch.arange(20, dtype=torch.float32).view(1, 1, 20, 1, 1)
        self.calls += 1
        return (base * (1 + ch)).contiguous()


def test_preprocess_and_plan(g7):
    assert glue.image_pre_processing(g7["frames"][:5], height=8).tobytes() == g7["pre5"].tobytes()
    for n in (17, 18, 33, 100, 2048):
        num, mode, starts = glue.sequence_plan(n)
        assert [num, mode] + list(starts) == g7[f"plan_{n}"].tolist()
    for fps in (25, 30):
        assert [glue.frame_offset_us(i, fps) for i in range(4096)] == g7[f"offsets_fps{fps}"].tolist()
    with pytest.raises(ValueError):
        glue.sequence_plan(16)


@pytest.mark.parametrize("infer_type", ["center", "pano"])
def test_video_to_voxels_matches_oracle_composition(g7, infer_type):
    H, WF, width, N, bs = g7["params"].tolist()
    frames = g7["frames"]
    got = glue.video_to_voxels(FakeModel(), frames=frames, infer_type=infer_type, seq_len=16,
                               width=width, height=H, batch_size=bs, device="cpu").numpy()
    # independent composition from the oracle's index arithmetic
    num, mode, starts = OG.sequence_plan(N)
    m = FakeModel()
    seq_out = {}
    for b0 in range(0, num, bs):
        units = np.stack([OG.preprocess(frames[starts[s]:starts[s] + 17]) for s in range(b0, min(b0 + bs, num))])
        x = torch.from_numpy(units)
        if infer_type == "center":
            lo, hi = OG.center_crop_cols(WF, width)
            out = m(x[..., lo:hi]).numpy()
        else:
            parts = []
            for lo, hi, keep in OG.pano_tiles(WF, width):
                o = m(x[..., lo:hi]).numpy()
                parts.append(o[..., -keep:] if keep else o)
            out = np.concatenate(parts, axis=-1)
        for j, s in enumerate(range(b0, min(b0 + bs, num))):
            seq_out[s] = out[j]
    want = np.stack([seq_out[s][j] for s, j in OG.merged_pair_sources(N)]).reshape(N - 1, 2, 10, H, -1)
    assert got.shape == want.shape == g7[infer_type].shape
    assert np.array_equal(got, want)


def test_resize_identity_and_shape():
    img = np.random.RandomState(0).rand(20, 30).astype(np.float32)
    assert glue._resize_bilinear(img, 30, 20) is img
    out = glue._resize_bilinear(img, 15, 10)
    assert out.shape == (10, 15) and abs(out.mean() - img.mean()) < 0.05


def test_resize_hand_vectors():
    """f3: glue._resize_bilinear against vectors derived BY HAND from OpenCV's scalar INTER_LINEAR
    (resize.cpp: scale = 1/(out/in); f = float((d + .5) * scale - .5); s = floor(f); f -= s; x axis zeroes f at
    both edges, y axis clips the rows; horizontal pass, then vertical); every product below is exact in f32.

    up, 2x2 -> 4x4, scale 0.5: f(d) = 0.5 d - 0.25 = -.25, .25, .75, 1.25
      x: d=0: s=-1 -> (f,s)=(0,0); d=1: s=0 f=.25; d=2: s=0 f=.75; d=3: s=1=W-1 -> copy S[1]
      y: d=0: s=-1, f=.75, rows (0,0); d=1: rows (0,1) f=.25; d=2: rows (0,1) f=.75; d=3: s=1 f=.25 rows (1,1)
      rows of [[0,1],[1,0]] after x: r0 = [0,.25,.75,1], r1 = [1,.75,.25,0]
      out: r0; .75 r0 + .25 r1; .25 r0 + .75 r1; r1
    down, 3x4 -> 2x2 of arange(12): x scale 2: f = .5 (s=0), 2.5 -> s=2 f=.5; y scale 1.5: f = .25 (s=0), 1.75 -> s=1 f=.75
      after x: row r = [4r + .5, 4r + 2.5]; out = [[.75*.5 + .25*4.5, .75*2.5 + .25*6.5], [.25*4.5 + .75*8.5, .25*6.5 + .75*10.5]]
    2x2 decimation (both scales exactly 2): OpenCV switches INTER_LINEAR to its INTER_AREA fast path, (a+b+c+d)/4."""
    up = glue._resize_bilinear(np.array([[0, 1], [1, 0]], np.float32), 4, 4)
    assert up.dtype == np.float32 and up.tolist() == [[0, .25, .75, 1], [.25, .375, .625, .75], [.75, .625, .375, .25], [1, .75, .25, 0]]
    down = glue._resize_bilinear(np.arange(12, dtype=np.float32).reshape(3, 4), 2, 2)
    assert down.tolist() == [[1.5, 3.5], [7.5, 9.5]]
    area = glue._resize_bilinear(np.arange(16, dtype=np.float32).reshape(4, 4), 2, 2)
    assert area.tolist() == [[2.5, 4.5], [10.5, 12.5]]
    # the float coefficient: (d + .5) * scale - .5 is rounded to f32 BEFORE the floor / subtraction (3 -> 7: scale 3/7)
    row = glue._resize_bilinear(np.array([[0, 1, 0]], np.float32), 7, 1)[0]
    sc = 1.0 / (7.0 / 3.0)
    want = []
    for d in range(7):
        f = np.float32((d + 0.5) * sc - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        src = [0.0, 1.0, 0.0]
        if s < 0:
            want.append(np.float32(src[0]))
        elif s >= 2:
            want.append(np.float32(src[2]))
        else:
            want.append(np.float32(src[s]) * (np.float32(1) - f) + np.float32(src[s + 1]) * f)
    assert row.tolist() == [float(v) for v in want]


def test_shard_range():
    for n in (1, 7, 8, 128):
        for world in (1, 2, 3, 8):
            got = [vdist.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("name", ["rgb", "gray", "rgb_ceil"])
def test_event_frames_match_reference(gold_dir, name):
    """Golden G9: the uint8 frames the reference's write_event_frame_video (v2ce.py:241-280) hands to
    cv2.VideoWriter (BGR), from the same voxel grid, through the product's reduction + image code."""
    from v2ce_toolbox_amd import pipeline, v2ce
    z = np.load(os.path.join(gold_dir, "event_frames_g9.npz"))
    keep, ceil, pct = (int(v) for v in z[f"args_{name}"])
    sums = pipeline.event_frame_sums(torch.from_numpy(z["vox"])).numpy()
    rgb = v2ce.event_frame_images(sums, ceil, pct, bool(keep))
    assert rgb.dtype == np.uint8
    assert rgb[..., ::-1].tobytes() == z[f"bgr_{name}"].tobytes()


def test_event_frame_writer_without_opencv(tmp_path, gold_dir):
    from v2ce_toolbox_amd import pipeline, v2ce
    z = np.load(os.path.join(gold_dir, "event_frames_g9.npz"))
    sums = pipeline.event_frame_sums(torch.from_numpy(z["vox"])).numpy()
    path = v2ce.write_event_frame_video(sums, str(tmp_path / "a-pred_ef_rgb.mp4"), 30, 10, 98, True)
    try:
        import cv2  # noqa: F401
        assert path.endswith(".mp4") and os.path.getsize(path) > 0
    except ImportError:
        assert path.endswith(".npz")
        assert np.load(path)["event_frames"][..., ::-1].tobytes() == z["bgr_rgb"].tobytes()


def test_streaming_npz_writer_and_sink(tmp_path):
    """f1: npz_stream.NpzStreamWriter writes the file np.savez(path, event_stream=...) writes (v2ce.py:371-372) as far as
    np.load is concerned, chunk by chunk; crc32_combine; the driver's streaming sink (CPU stand-in path) through v2ce.run."""
    import zipfile
    import zlib
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd import v2ce as cli
    from v2ce_toolbox_amd.LDATI import EVENT_DTYPE
    from v2ce_toolbox_amd.npz_stream import NpzStreamWriter, crc32_combine
    a, b = os.urandom(999), os.urandom(123457)
    assert crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    rng = np.random.default_rng(0)
    ev = np.zeros(50021, EVENT_DTYPE)
    ev["timestamp"], ev["x"] = np.sort(rng.integers(0, 1 << 40, len(ev))), rng.integers(0, 346, len(ev))
    ev["y"], ev["polarity"] = rng.integers(0, 260, len(ev)), rng.integers(0, 2, len(ev))
    p = str(tmp_path / "s.npz")
    with NpzStreamWriter(p, "event_stream", EVENT_DTYPE) as w:
        raw = ev.view(np.uint8)
        for i in range(0, len(raw), 13 * 4001):
            w.write(raw[i:i + 13 * 4001])
        with pytest.raises(ValueError):
            w.write(raw[:5])
    z = np.load(p)
    assert z.files == ["event_stream"] and z["event_stream"].dtype == EVENT_DTYPE and z["event_stream"].tobytes() == ev.tobytes()
    assert zipfile.ZipFile(p).testzip() is None
    with NpzStreamWriter(p, "event_stream", EVENT_DTYPE):
        pass
    assert np.load(p)["event_stream"].shape == (0,)
    # through the driver: the streamed file holds what the in-memory run returns
    sys_path_tests = os.path.dirname(os.path.abspath(__file__))
    import sys
    sys.path.insert(0, sys_path_tests)
    from test_dist_gloo import fake_stage2
    frames = synth.synthetic_frames(53, 8, 20, seed=3)
    kw = dict(infer_type="center", width=12, height=8, batch_size=2, device="cpu", stage2=fake_stage2(30))
    want = cli.run(frames, FakeModel(), **kw)
    n = cli.run(frames, FakeModel(), out_path=p, **kw)
    got = np.load(p)["event_stream"]
    assert n == len(want) == len(got) and got.tobytes() == want.tobytes()


def test_batch_shares_partition_the_clip():
    """pipeline.plan_batches / shard_of_batch (v2ce.py:149-154,211-239 as a plan): for any clip length, batch size and number
    of sequence shares the shares of a batch are disjoint, contiguous, in frame-pair order, and together keep exactly the
    frame-pairs the reference keeps (N - 1, the overlapped last sequence contributing its last `mode` pairs)."""
    from v2ce_toolbox_amd import pipeline
    for n_frames in (17, 18, 33, 49, 50, 100, 277, 2048):
        for bs in (1, 2, 4, 7, 32):
            plans = pipeline.plan_batches(n_frames, 16, bs)
            assert sum(p.n_pairs for p in plans) == n_frames - 1
            assert [p.first_pair for p in plans] == [16 * bs * i for i in range(len(plans))]
            for parts in (1, 2, 3, 4, 8):
                nxt = 0
                for bp in plans:
                    shares = [pipeline.shard_of_batch(bp, 16, r, parts) for r in range(parts)]
                    assert sum(len(s.seqs) for s in shares) == len(bp.seqs) and sum(s.n_pairs for s in shares) == bp.n_pairs
                    assert sum(1 for s in shares if s.drop) == (1 if bp.drop else 0)
                    for s in shares:
                        if s.seqs:
                            assert s.first_pair == nxt and s.starts == bp.starts[bp.seqs.index(s.seqs[0]):][:len(s.seqs)]
                            nxt += s.n_pairs if not s.drop else len(s.seqs) * 16      # pair indices count the dropped ones too
                    nxt = bp.first_pair + len(bp.seqs) * 16
