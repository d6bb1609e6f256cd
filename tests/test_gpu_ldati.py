"""GPU parity: HIP LDATI (through the C ABI) vs the C oracle and the reference goldens."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import ldati as O
from v2ce_toolbox_amd import hip, synth

pytestmark = pytest.mark.gpu


def hip_events(vox, fps=30, t0=0, uniforms=None, seed=None, frame_base=0, frame_ts_add=None, path="bucket",
               strategy="slope", layout="packed", **opts):
    from v2ce_toolbox_amd.LDATI import ldati_device
    y = torch.from_numpy(np.ascontiguousarray(vox)).cuda()
    u = None if uniforms is None else torch.from_numpy(np.ascontiguousarray(uniforms)).cuda()
    add = None if frame_ts_add is None else torch.from_numpy(frame_ts_add).cuda()
    ev = ldati_device(y, t0=t0, fps=fps, uniforms=u, seed=seed, frame_base=frame_base, frame_ts_add=add,
                      path=path, strategy=strategy, layout=layout, **opts)
    torch.cuda.synchronize()
    ev.check()
    return ev


def soa_equal(ev, seg, ts, x, y, p):
    assert np.array_equal(ev.seg_counts, seg)
    assert np.array_equal(ev.ts.cpu().numpy(), ts)
    assert np.array_equal(ev.x.cpu().numpy(), x)
    assert np.array_equal(ev.y.cpu().numpy(), y)
    assert np.array_equal(ev.p.cpu().numpy(), p)


@pytest.mark.parametrize("path", ["bucket", "sweep"])
@pytest.mark.parametrize("name", ["sparse", "frac", "stress", "t0fps60", "ragged", "none"])
def test_replay_matches_reference_golden_and_oracle(gold_dir, name, path):
    z = np.load(os.path.join(gold_dir, f"ldati_g3_{name}.npz"))
    vox, u, fps, t0 = z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"])
    ref = np.frombuffer(z["events"].tobytes(), O.EVENT_DTYPE)
    strategy = "none" if name == "none" else "slope"
    ev = hip_events(vox, fps, t0, uniforms=u, path=path, strategy=strategy)
    # bit-exact vs the oracle, including the stable tie order
    soa_equal(ev, *O.emit_soa(vox, fps=fps, t0=t0, uniforms=u, strategy=strategy))
    # vs the reference's own output: exact up to the tie order its unstable argsort leaves open
    mine = np.concatenate(ev.to_recarrays()) if ev.num_events else np.empty(0, O.EVENT_DTYPE)
    assert np.array_equal(ev.frame_counts, z["lens"])
    assert np.array_equal(mine["timestamp"], ref["timestamp"])
    assert O.canonicalize(mine, ev.seg_counts.reshape(-1)).tobytes() == \
        O.canonicalize(ref, ev.seg_counts.reshape(-1)).tobytes()


def test_low_fps_reference_golden(gold_dir):
    """G3 fps10: the reference's own output at 10 fps (11 111 us per bin: beyond the sweep kernel's LDS key histogram,
    so the two-level path alone serves it)."""
    z = np.load(os.path.join(gold_dir, "ldati_g3_fps10.npz"))
    vox, u, fps, t0 = z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"])
    assert hip.lib().v2ce_ldati_lds_bytes(fps, t0) == 0
    ref = np.frombuffer(z["events"].tobytes(), O.EVENT_DTYPE)
    ev = hip_events(vox, fps, t0, uniforms=u)
    soa_equal(ev, *O.emit_soa(vox, fps=fps, t0=t0, uniforms=u))
    mine = np.concatenate(ev.to_recarrays())
    assert np.array_equal(ev.frame_counts, z["lens"])
    assert np.array_equal(mine["timestamp"], ref["timestamp"])
    assert O.canonicalize(mine, ev.seg_counts.reshape(-1)).tobytes() == \
        O.canonicalize(ref, ev.seg_counts.reshape(-1)).tobytes()


OPTION_CASES = ["bidir", "bidir_sparse", "avg3", "avg5", "weighted", "random", "bidir_weighted"]


@pytest.mark.parametrize("layout", ["packed", "soa"])
@pytest.mark.parametrize("name", OPTION_CASES)
def test_option_goldens(gold_dir, name, layout):
    """bidirectional (LDATI.py:107-122), pooling (:177-182), 'random' (:173-174): the
    reference's own output for the same voxels and uniforms (REPLAY), and the oracle bit for bit."""
    z = np.load(os.path.join(gold_dir, f"ldati_g3_opt_{name}.npz"))
    vox, u, fps, t0 = z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"])
    ref = np.frombuffer(z["events"].tobytes(), O.EVENT_DTYPE)
    opts = dict(bidirectional=bool(z["bidirectional"]), pooling_type=str(z["pooling_type"]),
                pooling_kernel_size=int(z["pooling_kernel_size"]))
    strategy = str(z["strategy"])
    ev = hip_events(vox, fps, t0, uniforms=u, strategy=strategy, layout=layout, **opts)
    soa_equal(ev, *O.emit_soa(vox, fps=fps, t0=t0, uniforms=u, strategy=strategy, **opts))
    mine = np.concatenate(ev.to_recarrays())
    assert np.array_equal(ev.frame_counts, z["lens"])
    assert np.array_equal(mine["timestamp"], ref["timestamp"])
    assert O.canonicalize(mine, ev.seg_counts.reshape(-1)).tobytes() == \
        O.canonicalize(ref, ev.seg_counts.reshape(-1)).tobytes()


@pytest.mark.parametrize("opts", [dict(bidirectional=True), dict(pooling_type="weighted"),
                                  dict(pooling_type="avg", pooling_kernel_size=5), dict(strategy="random"),
                                  dict(bidirectional=True, pooling_type="avg")])
def test_options_philox_full_size(opts):
    """The options at 346x260 (several tiles per frame, halo of the pooling window across tile and
    image borders) against the oracle, Philox draws."""
    vox = synth.synthetic_voxels(2, 260, 346, seed=41, regime="sparse")
    strategy = opts.pop("strategy", "slope")
    want = O.emit_soa(vox, fps=30, seed=123, frame_base=2, strategy=strategy, **opts)
    soa_equal(hip_events(vox, seed=123, frame_base=2, strategy=strategy, **opts), *want)


def test_hand_kat(gold_dir):
    kat = json.load(open(os.path.join(gold_dir, "ldati_kat.json")))["hand"]
    vox = np.array(kat["vox"], np.float32).reshape(kat["shape"])
    u = np.array(kat["uniforms"], np.float32).reshape(kat["uniforms_shape"])
    from v2ce_toolbox_amd.LDATI import sample_voxel_statistical
    res = sample_voxel_statistical(torch.from_numpy(vox).cuda(), fps=30, uniforms=torch.from_numpy(u))
    assert [[int(v) for v in e] for e in res[0].tolist()] == kat["events"]
    assert res[0].dtype.itemsize == 13 and res[0].dtype.names == ("timestamp", "x", "y", "polarity")


@pytest.mark.parametrize("path", ["bucket", "sweep"])
def test_full_size_golden_sha(gold_dir, path):
    """G4: 346x260 dense frame; every segment >= 32768 events so the reference order is the stable
    order: packed bytes must be bit-identical to the reference's."""
    meta = json.load(open(os.path.join(gold_dir, "ldati_g4.json")))
    H, W = meta["H"], meta["W"]
    vox = synth.synthetic_voxels(1, H, W, seed=meta["vox_seed"], regime=meta["vox_regime"])
    mt = np.random.MT19937()
    mt._legacy_seeding(meta["torch_seed"])
    n = 2 * 9 * H * W * meta["max_n"]
    u = ((mt.random_raw(n).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24))
    u = u.reshape(1, 2, 9, H, W, meta["max_n"])
    ev = hip_events(vox, meta["fps"], meta["t0"], uniforms=u, path=path)
    assert ev.max_n == meta["max_n"]
    assert ev.seg_counts.reshape(-1).tolist() == meta["seg_counts"]
    rec = ev.to_recarrays()[0]
    assert hashlib.sha256(np.ascontiguousarray(rec["timestamp"]).tobytes()).hexdigest() == meta["sha256_timestamps"]
    assert hashlib.sha256(rec.tobytes()).hexdigest() == meta["sha256_packed_events"]


@pytest.mark.parametrize("shape,regime,fps,t0,fb", [
    ((3, 12, 14), "sparse", 30, 0, 0), ((2, 33, 47), "stress", 25, 0, 7),
    ((1, 1, 1), "stress", 30, 0, 0), ((2, 64, 80), "stress", 120, 0.25, 1000),
    ((1, 260, 346), "sparse", 30, 0, 5), ((4, 5, 129), "frac", 24, 0, 0)])
def test_philox_matches_oracle(shape, regime, fps, t0, fb):
    B, H, W = shape
    vox = synth.synthetic_voxels(B, H, W, seed=B * 1000 + H, regime=regime)
    want = O.emit_soa(vox, fps=fps, t0=t0, seed=0xDEADBEEF12345, frame_base=fb)
    for path, layout in (("bucket", "packed"), ("bucket", "soa"), ("sweep", "soa"), ("sweep", "packed")):
        ev = hip_events(vox, fps, t0, seed=0xDEADBEEF12345, frame_base=fb, path=path, layout=layout)
        soa_equal(ev, *want)


@pytest.mark.parametrize("shape,regime,fps", [((2, 40, 50), "stress", 10), ((1, 64, 64), "sparse", 5),
                                               ((2, 30, 41), "stress", 1)])
def test_low_fps_two_level_path(shape, regime, fps):
    """fps below ~12: a time bin spans more microsecond keys than the sweep kernel's LDS histogram
    holds (round 1 rejected these); the two-level path covers them (reference CLI accepts any fps)."""
    from v2ce_toolbox_amd import hip
    B, H, W = shape
    assert hip.lib().v2ce_ldati_lds_bytes(float(fps), 0.0) == 0
    vox = synth.synthetic_voxels(B, H, W, seed=fps, regime=regime)
    want = O.emit_soa(vox, fps=fps, seed=77, frame_base=3)
    for layout in ("packed", "soa"):
        soa_equal(hip_events(vox, fps, seed=77, frame_base=3, layout=layout), *want)


def test_pano_width_and_many_tiles():
    """1384-wide frames (BASELINE config 4: 352 tiles per frame) and a ragged last tile."""
    vox = synth.synthetic_voxels(1, 260, 1384, seed=3, regime="sparse")
    soa_equal(hip_events(vox, seed=11), *O.emit_soa(vox, seed=11))
    vox = synth.synthetic_voxels(2, 37, 167, seed=4, regime="stress")        # 6179 px: 3 full tiles + 35
    soa_equal(hip_events(vox, seed=12, frame_base=9), *O.emit_soa(vox, seed=12, frame_base=9))


def test_dense_tile_falls_back_to_sweep():
    """More events in one (tile, bin) than the tile pass holds in LDS: no two-level plan, the whole
    call takes the sweep kernel (bit-identical)."""
    vox = np.zeros((1, 2, 10, 50, 60), np.float32)
    vox[0, 0, 2] = 9.3                                            # 3000 px x 9.3 -> > 15360 events in a tile-bin
    vox[0, 1, 5, :7] = 1.7
    want = O.emit_soa(vox, seed=21)
    soa_equal(hip_events(vox, seed=21), *want)


def test_degenerate_ties_take_the_big_bucket_kernel():
    """Constant voxels: every pixel emits identical timestamps, so one (segment, key) bucket holds
    far more records than a sort workgroup's LDS; those buckets are ordered by the big-bucket kernel
    and the output is still the stable order."""
    vox = np.full((2, 2, 10, 120, 130), 0.5, np.float32)      # singles at identical times
    vox[1, :, 3] = 2.0                                        # plus multi-event voxels in one bin
    want = O.emit_soa(vox, fps=30, seed=9)
    assert want[0].max() > 8192
    for path in ("bucket", "sweep"):
        soa_equal(hip_events(vox, seed=9, path=path), *want)


def test_philox_batching_invariance():
    vox = synth.synthetic_voxels(5, 20, 30, seed=9, regime="stress")
    whole = hip_events(vox, seed=42, frame_base=100)
    a = hip_events(vox[:2], seed=42, frame_base=100)
    b = hip_events(vox[2:], seed=42, frame_base=102)
    assert np.array_equal(whole.ts.cpu().numpy(), np.concatenate([a.ts.cpu().numpy(), b.ts.cpu().numpy()]))
    assert np.array_equal(whole.x.cpu().numpy(), np.concatenate([a.x.cpu().numpy(), b.x.cpu().numpy()]))


def test_frame_offset_and_pack():
    vox = synth.synthetic_voxels(3, 16, 18, seed=4, regime="stress")
    add = np.array([0, 33333, 66666], np.int64)
    ev0 = hip_events(vox, seed=1)
    ev = hip_events(vox, seed=1, frame_ts_add=add)
    per = np.repeat(add, ev.frame_counts)
    assert np.array_equal(ev.ts.cpu().numpy(), ev0.ts.cpu().numpy() + per)
    rec = np.concatenate(ev.to_recarrays())
    want = O.pack(ev.ts.cpu().numpy(), ev.x.cpu().numpy(), ev.y.cpu().numpy(), ev.p.cpu().numpy())
    assert rec.tobytes() == np.asarray(want).tobytes()


def test_empty_input():
    from v2ce_toolbox_amd.LDATI import sample_voxel_statistical
    y = torch.zeros(2, 2, 10, 5, 6, device="cuda")
    res = sample_voxel_statistical(y)
    assert len(res) == 2 and all(len(r) == 0 and r.dtype.itemsize == 13 for r in res)
    with pytest.raises(RuntimeError):
        sample_voxel_statistical(y, strict_reference_errors=True)


def test_option_errors():
    from v2ce_toolbox_amd.LDATI import sample_voxel_statistical
    y = torch.zeros(1, 2, 10, 4, 4, device="cuda")
    with pytest.raises(AssertionError):
        sample_voxel_statistical(y, pooling_type="bogus")
    with pytest.raises(AssertionError):
        sample_voxel_statistical(y, additional_events_strategy="bogus")
    for kw in (dict(bidirectional=True), dict(additional_events_strategy="random"), dict(pooling_type="avg"),
               dict(additional_events_strategy="none")):
        assert len(sample_voxel_statistical(y, **kw)) == 1
    with pytest.raises(Exception):
        sample_voxel_statistical(torch.zeros(1, 2, 10, 4, 4))      # CPU tensor: no CPU path


def test_full_size_properties():
    """BASELINE chunk size (24 frame-pairs, 346x260): size-independent properties."""
    vox = synth.synthetic_voxels(24, 260, 346, seed=77, regime="sparse")
    ev = hip_events(vox, seed=5)
    seg, mx = O.count(vox[:2])
    assert np.array_equal(ev.seg_counts[:2], seg)
    ts = ev.ts.cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(ev.seg_counts.reshape(-1))])
    assert offs[-1] == ts.shape[0]
    d = np.diff(ts)
    bad = np.nonzero(d < 0)[0] + 1
    assert set(bad.tolist()) <= set(offs.tolist()), "timestamps must be sorted inside every segment"
    x, y, p = ev.x.cpu().numpy(), ev.y.cpu().numpy(), ev.p.cpu().numpy()
    assert x.min() >= 0 and x.max() <= 345 and y.min() >= 0 and y.max() <= 259 and set(np.unique(p)) <= {0, 1}
    # voxelising the events per (frame, polarity, bin) gives back the relocated counts: first frame
    n0 = int(ev.frame_counts[0])
    _, ots, ox, oy, op = O.emit_soa(vox[:1], seed=5)
    assert np.array_equal(ts[:n0], ots) and np.array_equal(x[:n0], ox) and np.array_equal(p[:n0], op)


def test_sparse_tile_kernel_equals_per_bin_kernel(monkeypatch):
    """The one-pass sparse tile kernel (tiles with <= 4096 events over the nine bins) and the per-bin
    tile kernel are two implementations of the same pass: same bytes with the sparse kernel disabled
    (V2CE_LDATI_NO_SPARSE), on frames that mix empty, sparse and dense tiles, and both equal the oracle."""
    rng = np.random.default_rng(21)
    H, W = 96, 160                                               # 8 tiles per polarity plane
    y = np.zeros((3, 2, 10, H, W), np.float32)
    y[:, :, :, :40] = np.maximum(0.25 * rng.standard_normal((3, 2, 10, 40, W)), 0).astype(np.float32)   # sparse rows
    y[:, :, :, 60:80] = (3.0 * rng.random((3, 2, 10, 20, W))).astype(np.float32)                        # dense rows
    # rows whose tiles hold about as many events as the sparse kernel's capacity (8192): both sides of the branch
    y[:, :, :, 40:60] = np.maximum(1.0 * rng.standard_normal((3, 2, 10, 20, W)), 0).astype(np.float32)
    want = O.emit_soa(y, fps=30, seed=9, frame_base=2)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("V2CE_LDATI_NO_SPARSE", "1")
        ev = hip_events(y, seed=9, frame_base=2)
        soa_equal(ev, *want)
        outs.append(ev.packed().cpu().numpy().tobytes())
    assert outs[0] == outs[1]


def rank_mode():
    import ctypes
    m = ctypes.c_int32(-1)
    hip.check(hip.lib().v2ce_ldati_rank_mode(ctypes.byref(m)), "v2ce_ldati_rank_mode")
    return m.value


def test_sort_workgroup_sizes_equal(gold_dir, monkeypatch):
    """The bucket sort runs workgroups of 128 threads (3072 records) on segments of real UNet output and of 256 (6144) on
    dense ones (make_plan's rule); V2CE_LDATI_SORT_THREADS forces 64 / 128 / 256: each size, on a sparse, a mid-density and a
    stress frame at full size and on the G4 stress frame, gives the bytes of the default rule, and G4 keeps the reference's SHA."""
    def all_sizes(run):
        monkeypatch.delenv("V2CE_LDATI_SORT_THREADS", raising=False)
        ref = run().packed().cpu().numpy().tobytes()
        for th in ("64", "128", "256"):
            monkeypatch.setenv("V2CE_LDATI_SORT_THREADS", th)
            assert run().packed().cpu().numpy().tobytes() == ref, th
        monkeypatch.delenv("V2CE_LDATI_SORT_THREADS")
        return ref

    for regime, scale in (("sparse", 1.0), ("frac", 1.0), ("stress", 1.0 / 3.0), ("stress", 1.0)):
        vox = (synth.synthetic_voxels(2, 260, 346, seed=91, regime=regime) * np.float32(scale)).astype(np.float32)
        all_sizes(lambda: hip_events(vox, seed=13))
    meta = json.load(open(os.path.join(gold_dir, "ldati_g4.json")))
    H, W = meta["H"], meta["W"]
    vox4 = synth.synthetic_voxels(1, H, W, seed=meta["vox_seed"], regime=meta["vox_regime"])
    mt = np.random.MT19937()
    mt._legacy_seeding(meta["torch_seed"])
    n = 2 * 9 * H * W * meta["max_n"]
    u4 = ((mt.random_raw(n).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24)).reshape(1, 2, 9, H, W, meta["max_n"])
    for th in ("128", "256"):
        monkeypatch.setenv("V2CE_LDATI_SORT_THREADS", th)
        ev = hip_events(vox4, meta["fps"], meta["t0"], uniforms=u4)
        assert hashlib.sha256(ev.to_recarrays()[0].tobytes()).hexdigest() == meta["sha256_packed_events"]
    monkeypatch.delenv("V2CE_LDATI_SORT_THREADS")


def test_ballot_rank_fallback_forced(gold_dir, monkeypatch):
    """The ballot match-any ranks -- what the tile pass, the bucket sort and the big-bucket kernel fall back to on a
    device whose LDS atomics fail the lane-order probe -- forced with V2CE_LDATI_NO_ATOMIC_ORDER=1 (the probe passes on
    gfx950, so the branch never runs otherwise): the reference goldens G3 (incl. bidirectional: the big-bucket kernel),
    G4 bit-identical, and a Philox full-size frame, each byte-equal to the default mode's output."""
    hip_events(synth.synthetic_voxels(1, 8, 8, seed=1), seed=1)          # the probe has run
    default_mode = rank_mode()
    print("LDS-atomic lane-order probe passed on this device:", bool(default_mode))

    def both(run):
        monkeypatch.delenv("V2CE_LDATI_NO_ATOMIC_ORDER", raising=False)
        a = run()
        monkeypatch.setenv("V2CE_LDATI_NO_ATOMIC_ORDER", "1")
        assert rank_mode() == 0
        b = run()
        monkeypatch.delenv("V2CE_LDATI_NO_ATOMIC_ORDER")
        assert rank_mode() == default_mode
        assert a.packed().cpu().numpy().tobytes() == b.packed().cpu().numpy().tobytes()
        return b

    for name in ["sparse", "frac", "stress", "ragged"]:
        z = np.load(os.path.join(gold_dir, f"ldati_g3_{name}.npz"))
        vox, u, fps, t0 = z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"])
        ev = both(lambda: hip_events(vox, fps, t0, uniforms=u))
        soa_equal(ev, *O.emit_soa(vox, fps=fps, t0=t0, uniforms=u))
    for name in ["bidir", "bidir_sparse"]:
        z = np.load(os.path.join(gold_dir, f"ldati_g3_opt_{name}.npz"))
        vox, u, fps, t0 = z["vox"], z["uniforms"], float(z["fps"]), float(z["t0"])
        ev = both(lambda: hip_events(vox, fps, t0, uniforms=u, bidirectional=True))
        soa_equal(ev, *O.emit_soa(vox, fps=fps, t0=t0, uniforms=u, bidirectional=True))
    meta = json.load(open(os.path.join(gold_dir, "ldati_g4.json")))
    H, W = meta["H"], meta["W"]
    vox4 = synth.synthetic_voxels(1, H, W, seed=meta["vox_seed"], regime=meta["vox_regime"])
    mt = np.random.MT19937()
    mt._legacy_seeding(meta["torch_seed"])
    n = 2 * 9 * H * W * meta["max_n"]
    u4 = ((mt.random_raw(n).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24)).reshape(1, 2, 9, H, W, meta["max_n"])
    ev = both(lambda: hip_events(vox4, meta["fps"], meta["t0"], uniforms=u4))
    assert hashlib.sha256(ev.to_recarrays()[0].tobytes()).hexdigest() == meta["sha256_packed_events"]
    for regime in ("sparse", "stress"):
        vox = synth.synthetic_voxels(2, 260, 346, seed=77, regime=regime)
        ev = both(lambda: hip_events(vox, seed=99, frame_base=4))
        if regime == "sparse":
            soa_equal(ev, *O.emit_soa(vox, fps=30, seed=99, frame_base=4))


def test_unphysical_voxels_are_refused():
    """A voxel grid with counts in the millions (broken checkpoint, un-normalised input) would ask for billions of events -- the
    reference dies allocating its dense [B,2,9,H,W,max_n] tensors (LDATI.py:171).  The build refuses before it allocates or
    launches anything of that size (a quick V2ceHipError; nothing unbounded reaches the device or the page-locked pool)."""
    import time
    vox = synth.synthetic_voxels(2, 32, 48, seed=3, regime="stress")
    vox[0, 0, 3, 5, 7] = 3.0e6
    t0 = time.perf_counter()
    with pytest.raises(hip.V2ceHipError, match="unphysical"):
        hip_events(vox, seed=1)
    assert time.perf_counter() - t0 < 5.0
    vox[0, 0, 3, 5, 7] = 4000.0                                  # large but within the limits: runs, bit-exact
    soa_equal(hip_events(vox, seed=1), *O.emit_soa(vox, fps=30, seed=1))


def test_exact_math_helpers():
    """The dense tile kernel's hand-scheduled square root / division (sqrt_rn_nr, div_rn_nr) and its bitop3 Philox rounds
    against the compiler's IEEE operations and the plain rounds: every slope-table entry x every Philox uniform
    (v2ce_ldati_selfcheck), at the CLI's fps and two others."""
    import ctypes
    for fps in (30.0, 25.0, 120.0):
        bad = (ctypes.c_int64 * 3)()
        hip.check(hip.lib().v2ce_ldati_selfcheck(fps, bad), "v2ce_ldati_selfcheck")
        assert (bad[0], bad[1], bad[2]) == (0, 0, 0), f"fps {fps}: {bad[0]} time mismatches, {bad[1]} Philox mismatches, {bad[2]} f64 single-time mismatches"


def test_dense_tile_kernel_equals_per_bin_kernel(gold_dir, monkeypatch):
    """ldati_tile_dense_kernel (round 4: the common call's dense tiles) and ldati_tile_pass_kernel (every other option,
    forced with V2CE_LDATI_OLD_TILE=1) are two implementations of one pass: same bytes on Philox stress frames (both
    workgroup sizes: the second case needs the 16-wave form), on mixed frames, on a voxel outside the slope table, with
    replayed uniforms (G4: the reference's own bytes), in 'none' mode and through the forced ballot ranks."""
    def both(run):
        monkeypatch.delenv("V2CE_LDATI_OLD_TILE", raising=False)
        a = run()
        monkeypatch.setenv("V2CE_LDATI_OLD_TILE", "1")
        b = run()
        monkeypatch.delenv("V2CE_LDATI_OLD_TILE")
        assert a.packed().cpu().numpy().tobytes() == b.packed().cpu().numpy().tobytes()
        assert np.array_equal(a.seg_counts, b.seg_counts)
        return a

    vox = synth.synthetic_voxels(2, 260, 346, seed=5, regime="stress")
    ev = both(lambda: hip_events(vox, seed=31, frame_base=3))
    seg, ts, x, y, p = O.emit_soa(vox[:1], fps=30, seed=31, frame_base=3)
    n0 = int(seg.sum())
    assert np.array_equal(ev.seg_counts[:1], seg) and np.array_equal(ev.ts.cpu().numpy()[:n0], ts)
    assert np.array_equal(ev.x.cpu().numpy()[:n0], x) and np.array_equal(ev.p.cpu().numpy()[:n0], p)
    dense = (10.0 * np.random.default_rng(3).random((1, 2, 10, 100, 160))).astype(np.float32)      # ~100 events per pixel
    soa_equal(both(lambda: hip_events(dense, seed=8)), *O.emit_soa(dense, fps=30, seed=8))
    # planes that are not 16-byte aligned (H W % 4 != 0): the dense kernel's general body with scalar plane loads
    odd = (8.0 * np.random.default_rng(4).random((1, 2, 10, 99, 161))).astype(np.float32)
    soa_equal(both(lambda: hip_events(odd, seed=9, frame_base=2)), *O.emit_soa(odd, fps=30, seed=9, frame_base=2))
    rng = np.random.default_rng(21)
    mix = np.zeros((2, 2, 10, 96, 160), np.float32)
    mix[:, :, :, :40] = np.maximum(0.25 * rng.standard_normal((2, 2, 10, 40, 160)), 0)
    mix[:, :, :, 40:] = 4.0 * rng.random((2, 2, 10, 56, 160))
    mix[0, 0, 4, 50, 7] = 45.0                                    # count 45 > 31: outside the slope table (owner-lane path)
    mix[1, 1, 2, 70, 9] = 60.0                                    # neighbours' difference > 31
    for fps, t0 in ((30, 0), (60, 0.5)):
        soa_equal(both(lambda: hip_events(mix, fps, t0, seed=77, frame_base=1)), *O.emit_soa(mix, fps=fps, t0=t0, seed=77, frame_base=1))
    soa_equal(both(lambda: hip_events(mix, seed=5, strategy="none")), *O.emit_soa(mix, fps=30, seed=5, strategy="none"))
    monkeypatch.setenv("V2CE_LDATI_NO_ATOMIC_ORDER", "1")
    soa_equal(both(lambda: hip_events(mix, seed=78)), *O.emit_soa(mix, fps=30, seed=78))
    monkeypatch.delenv("V2CE_LDATI_NO_ATOMIC_ORDER")
    # replayed uniforms: G4, the reference's own bytes
    meta = json.load(open(os.path.join(gold_dir, "ldati_g4.json")))
    H, W = meta["H"], meta["W"]
    vox4 = synth.synthetic_voxels(1, H, W, seed=meta["vox_seed"], regime=meta["vox_regime"])
    mt = np.random.MT19937()
    mt._legacy_seeding(meta["torch_seed"])
    n = 2 * 9 * H * W * meta["max_n"]
    u4 = ((mt.random_raw(n).astype(np.uint32) & 0xFFFFFF).astype(np.float32) * np.float32(2.0 ** -24)).reshape(1, 2, 9, H, W, meta["max_n"])
    ev = both(lambda: hip_events(vox4, meta["fps"], meta["t0"], uniforms=u4))
    assert hashlib.sha256(ev.to_recarrays()[0].tobytes()).hexdigest() == meta["sha256_packed_events"]


@pytest.mark.parametrize("form", ["V2CE_LDATI_PAIR", "V2CE_LDATI_ONEPASS"])
def test_pair_pass_kernel_equals_per_bin_kernel(form, monkeypatch):
    """ldati_tile_pair_kernel (round 6: two bins of a dense tile per pass where their records share the LDS; V2CE_LDATI_PAIR=1) and
    ldati_tile_onepass_kernel (the classification of all nine bins in one sweep, the work lists in global scratch, pair passes
    behind it; V2CE_LDATI_ONEPASS=1) against the per-bin dense kernel (the default): same bytes and the oracle's events -- on Philox stress frames (pairs 0-1 .. 6-7
    and bin 8 alone in the 16-wave form), at half and quarter that density (the 8-wave form, two workgroups per CU; at the
    lower density every pass is a pair), on a tile mix where pairs and single-bin passes alternate (a bin that fits with
    neither neighbour), with a voxel outside the slope table in either bin of a pair, at 60 fps with a start time, through the
    two-pass path (fused count off) and the dense slot mode (second call of a stream) alike."""
    from v2ce_toolbox_amd import LDATI

    def both(run):
        monkeypatch.setenv(form, "1")                          # (both forms are opt-in: DESIGN 4.2 round 6)
        a = run()
        monkeypatch.delenv(form)
        b = run()
        assert np.array_equal(a.seg_counts, b.seg_counts)
        assert a.packed().cpu().numpy().tobytes() == b.packed().cpu().numpy().tobytes()
        return a

    for scale, seed in ((6.0, 5), (3.0, 6), (1.5, 7)):
        vox = (scale * np.random.default_rng(seed).random((2, 2, 10, 260, 346))).astype(np.float32)
        LDATI._SEG_HINT.clear()
        ev = both(lambda: hip_events(vox, seed=31 + seed, frame_base=3))           # first call of a stream: the two-pass path
        ev2 = both(lambda: hip_events(vox, seed=31 + seed, frame_base=3))          # second call: the dense kernel is the count pass
        assert ev.packed().cpu().numpy().tobytes() == ev2.packed().cpu().numpy().tobytes()
        seg, ts, x, y, p = O.emit_soa(vox[:1], fps=30, seed=31 + seed, frame_base=3)
        n0 = int(seg.sum())
        assert np.array_equal(ev.seg_counts[:1], seg) and np.array_equal(ev.ts.cpu().numpy()[:n0], ts)
        assert np.array_equal(ev.x.cpu().numpy()[:n0], x) and np.array_equal(ev.y.cpu().numpy()[:n0], y)
        assert np.array_equal(ev.p.cpu().numpy()[:n0], p)
    # bins of very different weight: pairs where two light bins meet, single passes around the heavy ones
    rng = np.random.default_rng(11)
    mix = (rng.random((2, 2, 10, 128, 160)) * np.array([1, 9, 1, 1, 12, 12, 1, 0.2, 5, 3], np.float32)[None, None, :, None, None]).astype(np.float32)
    mix[0, 0, 4, 50, 7] = 45.0                                    # count 45 > 31: outside the slope table, second bin of a pair
    mix[1, 1, 2, 70, 9] = 60.0                                    # ... and the first bin of one
    mix[1, 0, :, :40] = 0.0                                       # an empty stretch: bins without a record inside a pass
    for fps, t0 in ((30, 0), (60, 0.5)):
        LDATI._SEG_HINT.clear()
        for _ in range(2):
            soa_equal(both(lambda: hip_events(mix, fps, t0, seed=77, frame_base=1)), *O.emit_soa(mix, fps=fps, t0=t0, seed=77, frame_base=1))
    # the fused count switched off: tile offsets from the count pass instead of slots
    monkeypatch.setenv("V2CE_LDATI_NO_FUSED", "1")
    soa_equal(both(lambda: hip_events(mix, seed=78)), *O.emit_soa(mix, fps=30, seed=78))
    monkeypatch.delenv("V2CE_LDATI_NO_FUSED")


def test_c5_stress_chunk_full_size(monkeypatch):
    """BASELINE config 5 at the size bench.py times: 24 frame-pairs of 346x260 `6 U[0,1)` voxels, Philox (127.7 M events:
    the fullest chunk histograms, 32-bit record offsets).  Counts of the first and the last frame vs the oracle, sortedness
    inside every segment, the first and last frame's events vs the oracle, one frame byte-equal with the sweep path, and
    the same chunk once through the forced ballot ranks."""
    vox = synth.synthetic_voxels(24, 260, 346, seed=7, regime="stress")
    y = torch.from_numpy(vox).cuda()
    from v2ce_toolbox_amd.LDATI import ldati_device

    def run():
        ev = ldati_device(y, fps=30, seed=0x5EED, frame_base=11)
        torch.cuda.synchronize()
        ev.check()
        return ev
    ev = run()
    assert ev.num_events > 120_000_000
    seg_first, _ = O.count(vox[:1])
    seg_last, _ = O.count(vox[-1:])
    assert np.array_equal(ev.seg_counts[:1], seg_first) and np.array_equal(ev.seg_counts[-1:], seg_last)
    ts = ev.ts
    offs = np.concatenate([[0], np.cumsum(ev.seg_counts.reshape(-1))])
    neg = (torch.nonzero(ts[1:] < ts[:-1]).flatten() + 1).cpu().numpy()
    assert set(neg.tolist()) <= set(offs.tolist()), "timestamps must be sorted inside every segment"
    x, yy, p = ev.x, ev.y, ev.p
    assert int(x.min()) >= 0 and int(x.max()) <= 345 and int(yy.min()) >= 0 and int(yy.max()) <= 259
    for f in (0, 23):
        lo, hi = int(ev.frame_counts[:f].sum()), int(ev.frame_counts[:f + 1].sum())
        seg, ots, ox, oy, op = O.emit_soa(vox[f:f + 1], fps=30, seed=0x5EED, frame_base=11 + f)
        assert np.array_equal(ts[lo:hi].cpu().numpy(), ots) and np.array_equal(x[lo:hi].cpu().numpy(), ox)
        assert np.array_equal(yy[lo:hi].cpu().numpy(), oy) and np.array_equal(p[lo:hi].cpu().numpy(), op)
    sweep = hip_events(vox[5:6], seed=0x5EED, frame_base=16, path="sweep")
    lo, hi = int(ev.frame_counts[:5].sum()), int(ev.frame_counts[:6].sum())
    assert sweep.packed().cpu().numpy().tobytes() == ev.packed()[13 * lo:13 * hi].cpu().numpy().tobytes()
    whole = ev.packed().cpu().numpy().tobytes()
    del ev, ts, x, yy, p
    monkeypatch.setenv("V2CE_LDATI_NO_ATOMIC_ORDER", "1")
    assert run().packed().cpu().numpy().tobytes() == whole


def test_fused_count_equals_two_pass(gold_dir, monkeypatch):
    """v2ce_ldati_count_fused (one pass over the voxels: counts + the sparse tiles' records in per-tile slots) against the
    two-pass path (V2CE_LDATI_NO_FUSED=1): same counts and same bytes on sparse full-size frames (the fused records are
    used), on frames with dense tiles (emit falls back by itself), at a low fps whose geometry differs, with replayed
    uniforms, 'none', bidirectional relocation and an event-free call."""
    def both(run):
        monkeypatch.delenv("V2CE_LDATI_NO_FUSED", raising=False)
        first = run()                                    # (the first call of a shape has no geometry hint and may fall back)
        a = run()
        assert first.packed().cpu().numpy().tobytes() == a.packed().cpu().numpy().tobytes()
        monkeypatch.setenv("V2CE_LDATI_NO_FUSED", "1")
        b = run()
        monkeypatch.delenv("V2CE_LDATI_NO_FUSED")
        assert np.array_equal(a.seg_counts, b.seg_counts) and a.max_n == b.max_n
        assert a.packed().cpu().numpy().tobytes() == b.packed().cpu().numpy().tobytes()
        return a

    vox = synth.synthetic_voxels(3, 260, 346, seed=12, regime="sparse")
    soa_equal(both(lambda: hip_events(vox, seed=3, frame_base=9)), *O.emit_soa(vox, fps=30, seed=3, frame_base=9))
    add = np.array([5, 33338, 66671], np.int64)
    both(lambda: hip_events(vox, seed=3, frame_base=9, frame_ts_add=add, layout="soa"))
    rng = np.random.default_rng(2)
    mix = np.zeros((2, 2, 10, 96, 160), np.float32)
    mix[:, :, :, :48] = np.maximum(0.3 * rng.standard_normal((2, 2, 10, 48, 160)), 0)
    mix[1, :, :, 48:] = 3.0 * rng.random((2, 10, 48, 160))                  # dense tiles in the second frame only
    soa_equal(both(lambda: hip_events(mix, seed=4)), *O.emit_soa(mix, fps=30, seed=4))
    small = synth.synthetic_voxels(2, 40, 50, seed=10, regime="stress")
    soa_equal(both(lambda: hip_events(small, 10, seed=77, frame_base=3)), *O.emit_soa(small, fps=10, seed=77, frame_base=3))
    soa_equal(both(lambda: hip_events(vox[:1], seed=6, strategy="none")), *O.emit_soa(vox[:1], fps=30, seed=6, strategy="none"))
    soa_equal(both(lambda: hip_events(vox[:1], seed=6, bidirectional=True)), *O.emit_soa(vox[:1], fps=30, seed=6, bidirectional=True))
    z = np.load(os.path.join(gold_dir, "ldati_g3_ragged.npz"))
    soa_equal(both(lambda: hip_events(z["vox"], float(z["fps"]), float(z["t0"]), uniforms=z["uniforms"])),
              *O.emit_soa(z["vox"], fps=float(z["fps"]), t0=float(z["t0"]), uniforms=z["uniforms"]))
    assert both(lambda: hip_events(np.zeros((2, 2, 10, 30, 40), np.float32), seed=1)).num_events == 0


def test_c_abi_fused_dense_call_sequence(monkeypatch, capfd):
    """The dense form of the fused count (expected_max_tile_bin_events > 0: ldati_tile_dense_kernel is the count pass and the tile
    pass at once, a slot per (tile, bin)) through ctypes alone: a first call without expectations (the sparse form; its tiles
    are dense, the emit takes the two-pass path), then with the first call's statistics as expectations (the slots are used:
    checked in the library's debug line), then with an expectation that is too small (the emit repeats the tile pass).  All
    three give the oracle's events; the statistics of the dense form equal the count pass's."""
    import ctypes
    L = hip.lib()
    rng = np.random.default_rng(5)
    vox = (5.0 * rng.random((2, 2, 10, 96, 160))).astype(np.float32)
    vox[1, :, :, :40] = np.maximum(0.3 * rng.standard_normal((2, 10, 40, 160)), 0)        # sparse tiles too: the dense kernel takes them all
    y = torch.from_numpy(vox).cuda()
    B, _, _, H, W = y.shape
    want = O.emit_soa(vox, fps=30, seed=42, frame_base=2)
    st = hip.stream_ptr(y.device)
    o = hip.LdatiOptions(hip.STRATEGY_SLOPE, 0, hip.POOL_NONE, 3)
    monkeypatch.setenv("V2CE_LDATI_DEBUG", "1")
    hint, bin_hint, ref_stats = 0, 0, None
    for attempt, expect_fused in ((0, False), (1, True), (2, False)):
        if attempt == 2:
            bin_hint = max(256, ref_stats[1] // 2)
        fb = L.v2ce_ldati_fused_ws_bytes(B, H, W, 30.0, 0.0, ctypes.byref(o), hint, bin_hint)
        assert fb > 0
        tws = torch.empty(L.v2ce_ldati_tile_ws_bytes(B, H, W), dtype=torch.uint8, device="cuda")
        fws = torch.empty(fb, dtype=torch.uint8, device="cuda")
        meta = torch.empty(B * 9 + 1 + 8, dtype=torch.int64, device="cuda")
        hip.check(L.v2ce_ldati_count_fused(y.data_ptr(), B, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 42, 2, hint, bin_hint,
                                           tws.data_ptr(), tws.numel(), fws.data_ptr(), fws.numel(), meta.data_ptr(),
                                           meta[B * 9 + 1:].data_ptr(), st), "count_fused")
        host = meta.cpu().numpy()
        max_n, max_tile, max_seg, total, tile_all = (int(v) for v in host[B * 9 + 1:B * 9 + 6])
        assert total == int(want[0].sum()) and np.array_equal(np.diff(host[:B * 9 + 1]).reshape(B, 9), want[0]) and tile_all > 8192
        if ref_stats is None:
            ref_stats = (max_n, max_tile, max_seg, total, tile_all)
        assert (max_n, max_tile, max_seg, total, tile_all) == ref_stats          # the dense kernel counts what the count pass counts
        nb = L.v2ce_ldati_workspace_bytes(B, H, W, 30.0, 0.0, ctypes.byref(o), total, max_seg, max_tile, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        out = torch.empty(total * 13 + 64, dtype=torch.uint8, device="cuda")
        capfd.readouterr()
        hip.check(L.v2ce_ldati_emit_fused(y.data_ptr(), B, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 42, 2,
                                          meta.data_ptr(), None, None, None, None, None, out.data_ptr(), total, max_seg, max_tile,
                                          tws.data_ptr(), ws.data_ptr(), nb, fws.data_ptr(), fws.numel(), tile_all, hint, bin_hint, st), "emit_fused")
        torch.cuda.synchronize()
        err = capfd.readouterr().err
        assert ("fused=1" in err) == expect_fused, err
        rec = out[:total * 13].cpu().numpy().view(O.EVENT_DTYPE)
        assert np.array_equal(rec["timestamp"], want[1]) and np.array_equal(rec["x"], want[2])
        assert np.array_equal(rec["y"], want[3]) and np.array_equal(rec["polarity"], want[4])
        hint, bin_hint = max_seg, max_tile + 64


def test_dense_slot_mode_counts_a_tile_bin_beyond_16_bits():
    """in slot mode the dense tile kernel is the count pass; its packed 16 + 16-bit wave totals wrapped for a
    (tile, bin) of >= 65536 events (mean >= 32 per pixel and bin: unphysical, but reachable right behind a dense call that armed
    the dense hint), the wrapped count passed the slot test and the emit accepted garbage.  A dense call, then a grid of ~40
    events per voxel of the same shape: counts, statistics and events equal the two-pass path's / the oracle's."""
    import ctypes
    from v2ce_toolbox_amd import LDATI as LD
    L = hip.lib()
    rng = np.random.default_rng(11)
    H, W = 64, 128                                              # 4 tiles of 2048 pixels per polarity plane
    dense = (5.0 * rng.random((1, 2, 10, H, W))).astype(np.float32)
    huge = (36.0 + 8.0 * rng.random((1, 2, 10, H, W))).astype(np.float32)        # ~40 events per voxel: ~82 k per (tile, bin)
    LD._SEG_HINT.clear()
    for vox in (dense, dense, huge):                            # (the second call runs the dense fused form; the third inherits its hint)
        y = torch.from_numpy(vox).cuda()
        ev = LD.ldati_device(y, fps=30, seed=9)
        want = O.emit_soa(vox, fps=30, seed=9)
        assert np.array_equal(ev.seg_counts, want[0])
        assert np.array_equal(ev.ts.cpu().numpy(), want[1]) and np.array_equal(ev.x.cpu().numpy(), want[2])
        assert np.array_equal(ev.y.cpu().numpy(), want[3]) and np.array_equal(ev.p.cpu().numpy(), want[4])
    # and at the ABI: the dense fused count with a slot hint reports the true (unwrapped) statistics of the huge grid
    y = torch.from_numpy(huge).cuda()
    o = hip.LdatiOptions(hip.STRATEGY_SLOPE, 0, hip.POOL_NONE, 3)
    st = hip.stream_ptr(y.device)
    tws = torch.empty(L.v2ce_ldati_tile_ws_bytes(1, H, W), dtype=torch.uint8, device="cuda")
    meta = torch.empty(9 + 1 + 8, dtype=torch.int64, device="cuda")
    hip.check(L.v2ce_ldati_count(y.data_ptr(), 1, H, W, ctypes.byref(o), tws.data_ptr(), tws.numel(), meta.data_ptr(), meta[10:].data_ptr(), st), "count")
    ref = meta.cpu().numpy().copy()
    seg_hint, bin_hint = int(ref[12]), 12000
    fb = L.v2ce_ldati_fused_ws_bytes(1, H, W, 30.0, 0.0, ctypes.byref(o), seg_hint, bin_hint)
    assert fb > 0
    fws = torch.empty(fb, dtype=torch.uint8, device="cuda")
    meta2 = torch.zeros(9 + 1 + 8, dtype=torch.int64, device="cuda")
    hip.check(L.v2ce_ldati_count_fused(y.data_ptr(), 1, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 9, 0, seg_hint, bin_hint,
                                       tws.data_ptr(), tws.numel(), fws.data_ptr(), fws.numel(), meta2.data_ptr(), meta2[10:].data_ptr(), st), "count_fused")
    got = meta2.cpu().numpy()
    assert np.array_equal(got[:10], ref[:10]), (got[:10], ref[:10])                       # segment offsets
    assert np.array_equal(got[10:14], ref[10:14]) and got[11] > 65535, (got[10:15], ref[10:14])   # max_n, max (tile, bin), max segment, total


def test_fused_dense_form_falls_back_for_calls_the_dense_kernel_does_not_serve():
    """Dense voxels at t0 = 3000 s need 64-bit times: the dense tile kernel does not serve such a call.  With the expectations of
    a dense stream ldati_begin still asks for the dense form of the fused count; the library then runs the plain count pass
    inside it and the emit takes the two-pass path (per-bin kernel).  Three consecutive calls (no history / dense expectation /
    whatever the second left behind) all give the oracle's events; the same at fps 10 with bidirectional relocation (no
    dense form at all: v2ce_ldati_fused_ws_bytes = 0)."""
    rng = np.random.default_rng(17)
    vox = (5.0 * rng.random((2, 2, 10, 64, 160))).astype(np.float32)
    want = O.emit_soa(vox, fps=30, t0=3000.0, seed=21, frame_base=1)
    for _ in range(3):
        soa_equal(hip_events(vox, 30, 3000.0, seed=21, frame_base=1), *want)
    want = O.emit_soa(vox, fps=10, seed=22, bidirectional=True)
    for _ in range(3):
        soa_equal(hip_events(vox, 10, 0, seed=22, bidirectional=True), *want)


def test_c_abi_fused_call_sequence():
    """The call sequence INTEGRATION.md gives a maintainer of the reference, through ctypes alone (no LDATI.py): hint 0 ->
    v2ce_ldati_fused_ws_bytes -> v2ce_ldati_count_fused -> read stats -> v2ce_ldati_workspace_bytes -> v2ce_ldati_emit_fused;
    once with the default hint (the geometry may miss: the emit falls back by itself) and once with the real densest-segment
    count as the hint.  Both give the oracle's events; argument errors come back as codes, never as exceptions."""
    import ctypes
    L = hip.lib()
    vox = synth.synthetic_voxels(2, 96, 160, seed=31, regime="sparse")
    y = torch.from_numpy(vox).cuda()
    B, _, _, H, W = y.shape
    want = O.emit_soa(vox, fps=30, seed=99, frame_base=5)
    st = hip.stream_ptr(y.device)
    o = hip.LdatiOptions(hip.STRATEGY_SLOPE, 0, hip.POOL_NONE, 3)
    hint = 0
    for attempt in range(2):
        fb = L.v2ce_ldati_fused_ws_bytes(B, H, W, 30.0, 0.0, ctypes.byref(o), hint, 0)
        assert fb > 0
        tws = torch.empty(L.v2ce_ldati_tile_ws_bytes(B, H, W), dtype=torch.uint8, device="cuda")
        fws = torch.empty(fb, dtype=torch.uint8, device="cuda")
        meta = torch.empty(B * 9 + 1 + 8, dtype=torch.int64, device="cuda")
        hip.check(L.v2ce_ldati_count_fused(y.data_ptr(), B, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 99, 5, hint, 0,
                                           tws.data_ptr(), tws.numel(), fws.data_ptr(), fws.numel(), meta.data_ptr(),
                                           meta[B * 9 + 1:].data_ptr(), st), "count_fused")
        host = meta.cpu().numpy()
        max_n, max_tile, max_seg, total, tile_all = (int(v) for v in host[B * 9 + 1:B * 9 + 6])
        assert total == int(want[0].sum()) and np.array_equal(np.diff(host[:B * 9 + 1]).reshape(B, 9), want[0]) and tile_all <= 8192
        nb = L.v2ce_ldati_workspace_bytes(B, H, W, 30.0, 0.0, ctypes.byref(o), total, max_seg, max_tile, 1)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        out = torch.empty(total * 13 + 64, dtype=torch.uint8, device="cuda")
        hip.check(L.v2ce_ldati_emit_fused(y.data_ptr(), B, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 99, 5,
                                          meta.data_ptr(), None, None, None, None, None, out.data_ptr(), total, max_seg, max_tile,
                                          tws.data_ptr(), ws.data_ptr(), nb, fws.data_ptr(), fws.numel(), tile_all, hint, 0, st), "emit_fused")
        rec = out[:total * 13].cpu().numpy().view(O.EVENT_DTYPE)
        assert np.array_equal(rec["timestamp"], want[1]) and np.array_equal(rec["x"], want[2])
        assert np.array_equal(rec["y"], want[3]) and np.array_equal(rec["polarity"], want[4])
        hint = max_seg
    # errors are return codes with a message
    assert L.v2ce_ldati_count_fused(y.data_ptr(), B, H, W, 30.0, 0.0, ctypes.byref(o), hip.RNG_PHILOX, None, 0, 99, 5, 0, 0, tws.data_ptr(),
                                    tws.numel(), fws.data_ptr(), 16, meta.data_ptr(), meta[B * 9 + 1:].data_ptr(), st) != 0
    assert b"workspace" in L.v2ce_last_error()
    rnd = hip.LdatiOptions(hip.STRATEGY_RANDOM, 0, hip.POOL_NONE, 3)
    assert L.v2ce_ldati_fused_ws_bytes(B, H, W, 30.0, 0.0, ctypes.byref(rnd), 0, 0) == 0        # 'random' has no fused path
    assert L.v2ce_ldati_fused_ws_bytes(B, H, W, 30.0, 0.0, ctypes.byref(rnd), 0, 5000) == 0
