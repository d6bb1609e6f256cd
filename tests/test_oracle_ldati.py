"""CPU: the C LDATI oracle against the golden vectors captured from the reference
(scripts/LDATI.py run on CPU by oracle/make_goldens.py) and the known answers."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import ldati as O
from v2ce_toolbox_amd import synth

CASES = ["sparse", "frac", "stress", "t0fps60", "ragged", "none", "fps10"]


def load_g3(gold_dir, name):
    z = np.load(os.path.join(gold_dir, f"ldati_g3_{name}.npz"))
    ev = np.frombuffer(z["events"].tobytes(), O.EVENT_DTYPE)
    return z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"]), z["lens"], ev


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(gold_dir, name):
    vox, u, fps, t0, lens, ref = load_g3(gold_dir, name)
    strategy = "none" if name == "none" else "slope"       # additional_events_strategy
    seg, ts, x, y, p = O.emit_soa(vox, fps=fps, t0=t0, uniforms=u, strategy=strategy)
    mine = np.asarray(O.pack(ts, x, y, p))
    assert np.array_equal(seg.sum(axis=1), lens)            # per-frame counts exact
    assert mine.dtype.itemsize == 13
    # segments < 32768 events: reference argsort is unstable -> compare canonicalised ties
    a = O.canonicalize(ref, seg.reshape(-1))
    b = O.canonicalize(mine, seg.reshape(-1))
    assert a.tobytes() == b.tobytes()
    # timestamps are identical position by position even without canonicalisation
    assert np.array_equal(ref["timestamp"], mine["timestamp"])


OPTION_CASES = ["bidir", "bidir_sparse", "avg3", "avg5", "weighted", "random", "bidir_weighted"]


def load_opt(gold_dir, name):
    z = np.load(os.path.join(gold_dir, f"ldati_g3_opt_{name}.npz"))
    ev = np.frombuffer(z["events"].tobytes(), O.EVENT_DTYPE)
    opts = dict(strategy=str(z["strategy"]), bidirectional=bool(z["bidirectional"]),
                pooling_type=str(z["pooling_type"]), pooling_kernel_size=int(z["pooling_kernel_size"]))
    return z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"]), z["lens"], ev, opts


@pytest.mark.parametrize("name", OPTION_CASES)
def test_oracle_matches_reference_option_goldens(gold_dir, name):
    """bidirectional (LDATI.py:107-122), pooling (:177-182), 'random' (:173-174) against the
    reference's own output for the same voxels and uniforms."""
    vox, u, fps, t0, lens, ref, opts = load_opt(gold_dir, name)
    seg, ts, x, y, p = O.emit_soa(vox, fps=fps, t0=t0, uniforms=u, **opts)
    mine = np.asarray(O.pack(ts, x, y, p))
    assert np.array_equal(seg.sum(axis=1), lens)
    assert np.array_equal(ref["timestamp"], mine["timestamp"])
    a = O.canonicalize(ref, seg.reshape(-1))
    b = O.canonicalize(mine, seg.reshape(-1))
    assert a.tobytes() == b.tobytes()


def test_bidirectional_relocation_quirks():
    """bin 4 is never written by the reference's bidirectional branch (stays 0); bin 8's tendency is y[9]."""
    y = np.array([3.6122, 5.5396, 1.5710, 5.0188, 1.3165, 2.8533, 1.7283, 1.4550, 5.8432, 2.1986], np.float32)
    n, tend = O.relocate2(y, bidirectional=True)
    assert n.tolist() == [4, 6, 1, 5, 0, 3, 2, 1, 8]
    assert tend[4] == 0 and tend[8] == y[9] and tend[5] < 0


def test_hand_kat(gold_dir):
    kat = json.load(open(os.path.join(gold_dir, "ldati_kat.json")))["hand"]
    vox = np.array(kat["vox"], np.float32).reshape(kat["shape"])
    u = np.array(kat["uniforms"], np.float32).reshape(kat["uniforms_shape"])
    seg, mx = O.count(vox)
    assert mx == 4 and int(seg.sum()) == 23
    n0, d0 = O.relocate(vox[0, 0, :, 0, 0])
    assert n0.tolist() == [1, 0, 1, 0, 0, 1, 0, 0, 1]
    assert np.allclose(d0, [.7, .3, .8, .8, .8, .6, .6, .6, .7], atol=1e-6)
    assert O.relocate(vox[0, 0, :, 0, 1])[0].tolist() == [2, 4, 0, 0, 4, 1, 0, 0, 2]
    assert O.relocate(vox[0, 1, :, 0, 0])[0].tolist() == [0, 0, 1, 1, 1, 0, 0, 3, 0]
    ev = O.sample_voxel_statistical_oracle(vox, fps=30, uniforms=u)[0]
    got = [[int(v) for v in e] for e in ev.tolist()]
    assert got == kat["events"]
    # SURVEY 8c literal list (first / deterministic ones)
    assert got[0] == [1138, 1, 0, 1] and got[2] == [2592, 0, 0, 1] and got[-1] == [33259, 1, 0, 1]
    assert int(0.7 * (1e6 / 270)) == 2592


def test_notebook_kat(gold_dir):
    """train/scripts/stage2/vis_stage2.ipynb cell 2: the three single-event times are deterministic."""
    kat = json.load(open(os.path.join(gold_dir, "ldati_kat.json")))["notebook"]
    vox = np.array(kat["vox"], np.float32).reshape(1, 2, 10, 1, 1)
    ev = O.sample_voxel_statistical_oracle(vox, fps=30, seed=3)[0]
    ts = np.sort(ev["timestamp"]) * (300 / 1e6)
    assert len(ts) == kat["num_events"] == 8
    assert np.allclose(ts[:3], kat["printed_first_three"], atol=5e-5)
    assert np.allclose(ts[:3], kat["reference_first_three_here"], atol=0)


def test_offsets_match_torch_arange():
    """The f32 bin offsets are torch.arange(0, 1/fps, 1/fps/9) (LDATI.py:163)."""
    import torch
    for fps in (24, 25, 30, 50, 60, 120, 240):
        vs = 1 / fps / 9
        a = torch.arange(0, 1 / fps, vs).numpy()
        assert a.shape == (9,)
        mine = np.array([np.float32(c * vs) for c in range(9)], np.float32)
        assert a.tobytes() == mine.tobytes()
        O.check_arange_len(fps)
    assert np.array([np.float32(c / 270) for c in range(9)]).view(np.uint32).tolist()[1:4] == \
        [997374422, 1005763030, 1010174817]


def test_mt19937_identity():
    """G6: CPU torch.rand == (MT19937 raw & 0xFFFFFF) * 2^-24 in linear order."""
    import torch
    torch.manual_seed(99)
    a = torch.rand(3, 5, 7, 11).numpy()
    mt = np.random.MT19937()
    mt._legacy_seeding(99)
    b = ((mt.random_raw(a.size).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24))
    assert np.array_equal(a.reshape(-1), b)


def test_large_golden_sha(gold_dir):
    """G4: full-size dense frame, every segment >= 32768 events => reference order is the stable
    order and the packed bytes must match the reference bit for bit."""
    meta = json.load(open(os.path.join(gold_dir, "ldati_g4.json")))
    H, W = meta["H"], meta["W"]
    vox = synth.synthetic_voxels(1, H, W, seed=meta["vox_seed"], regime=meta["vox_regime"])
    mt = np.random.MT19937()
    mt._legacy_seeding(meta["torch_seed"])
    n = 2 * 9 * H * W * meta["max_n"]
    u = ((mt.random_raw(n).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24))
    u = u.reshape(1, 2, 9, H, W, meta["max_n"])
    seg, ts, x, y, p = O.emit_soa(vox, fps=meta["fps"], t0=meta["t0"], uniforms=u)
    assert seg.reshape(-1).tolist() == meta["seg_counts"]
    ev = np.asarray(O.pack(ts, x, y, p))
    assert hashlib.sha256(np.ascontiguousarray(ev["timestamp"]).tobytes()).hexdigest() == meta["sha256_timestamps"]
    assert hashlib.sha256(ev.tobytes()).hexdigest() == meta["sha256_packed_events"]


def test_philox_fill_consistent():
    u = O.philox_uniforms(2, 3, 4, 5, seed=77, frame_base=10)
    assert u.shape == (2, 2, 9, 3, 4, 5) and u.min() >= 0 and u.max() < 1
    v = O.lib().v2ce_oracle_philox_uniform(77, 2 * 4 + 3, 4, 1 * 9 + 5, 11)
    assert u[1, 1, 5, 2, 3, 4] == v
    # Philox4x32-10 known-answer (Random123 kat_vectors: counter=0, key=0)
    import ctypes
    w0 = O.lib().v2ce_oracle_philox_uniform(0, 0, 0, 0, 0)
    assert w0 == np.float32((0x6627e8d5 >> 8) * 2.0 ** -24)


def test_replay_equals_philox_when_fed_philox_uniforms():
    vox = synth.synthetic_voxels(2, 6, 7, seed=5, regime="stress")
    _, mx = O.count(vox)
    u = O.philox_uniforms(2, 6, 7, mx, seed=1234, frame_base=3)
    a = O.emit_soa(vox, uniforms=u)
    b = O.emit_soa(vox, seed=1234, frame_base=3)
    for s, t in zip(a, b):
        assert np.array_equal(s, t)
