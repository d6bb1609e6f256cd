"""GPU: the ablation samplers (csrc/sampler.hip through the C ABI) against the reference goldens G10
and against the oracle on seeded inputs, replayed draws and Philox draws; bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import sample_methods as OS
from v2ce_toolbox_amd import hip, sample_methods as SM, synth

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[len("sampler_g10_"):-4]
               for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sampler_g10_*.npz")))


def run_device(vox, kind, mode, t0, fps, **kw):
    y = torch.from_numpy(np.ascontiguousarray(vox)).cuda()
    kw = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    if kind == "baseline":
        res = SM.sample_voxel_baseline(y, t0, fps, even=mode == "even", random=mode == "random", **kw)
    else:
        res = SM.sample_voxel_pure_slope(y, t0, fps, **kw)
    assert torch.equal(y.cpu(), torch.from_numpy(vox))          # the input is never modified
    return res


def run_oracle(vox, kind, mode, t0, fps, **kw):
    if kind == "baseline":
        return OS.sample_voxel_baseline(vox, t0, fps, even=mode == "even", random=mode == "random", **kw)
    return OS.sample_voxel_pure_slope(vox, t0, fps, **kw)


def same(a, b):
    assert [len(r) for r in a] == [len(r) for r in b]
    for i, (ra, rb) in enumerate(zip(a, b)):
        assert np.asarray(ra).dtype.itemsize == 13
        assert np.asarray(ra).tobytes() == np.asarray(rb).tobytes(), f"frame {i}"


@pytest.mark.parametrize("name", CASES)
def test_reference_goldens(gold_dir, name):
    z = np.load(os.path.join(gold_dir, f"sampler_g10_{name}.npz"))
    res = run_device(z["vox"], str(z["kind"]), str(z["mode"]), float(z["t0"]), float(z["fps"]),
                     u_int=z["u_int"], u_dec=z["u_dec"], u_bern=z["u_bern"])
    assert [len(r) for r in res] == z["lens"].tolist()
    assert np.concatenate([np.asarray(r) for r in res]).tobytes() == z["events"].tobytes()


@pytest.mark.parametrize("kind,mode", [("baseline", "random"), ("baseline", "even"), ("pure_slope", "slope")])
@pytest.mark.parametrize("regime,shape,fps,t0", [("stress", (2, 13, 37), 30, 0), ("sparse", (3, 20, 70), 25, 1.5),
                                                 ("frac", (1, 1, 129), 120, 0), ("stress", (1, 67, 1), 50, 0.125)])
def test_philox_vs_oracle(kind, mode, regime, shape, fps, t0):
    """Philox mode == the oracle fed with the materialised Philox draws (same counters), incl. a non-zero frame base."""
    B, H, W = shape
    vox = synth.synthetic_voxels(B, H, W, seed=B * 100 + H, regime=regime)
    M = int(np.floor(vox[:, :, :8]).max(initial=0))
    M = max(M, int(np.floor(vox[:, :, 8] + vox[:, :, 9]).max(initial=0)), int(np.floor(vox).max(initial=0)))
    u_int, u_dec, u_bern = OS.philox_draws(B, H, W, M, seed=77, frame_base=5)
    want = run_oracle(vox, kind, mode, t0, fps, u_int=u_int, u_dec=u_dec, u_bern=u_bern)
    got = run_device(vox, kind, mode, t0, fps, seed=77, frame_base=5)
    same(got, want)
    assert sum(len(r) for r in got) > 0


def test_batching_invariance_and_determinism():
    """Philox draws are keyed by the global frame index: a batch split in two gives the same frames; two runs of the
    same call give the same bytes (the unordered key writes never reach the output)."""
    vox = synth.synthetic_voxels(4, 24, 40, seed=3, regime="stress")
    y = torch.from_numpy(vox).cuda()
    for fn in (lambda v, fb: SM.sample_voxel_baseline(v, random=True, seed=9, frame_base=fb),
               lambda v, fb: SM.sample_voxel_pure_slope(v, seed=9, frame_base=fb)):
        whole = fn(y, 0)
        again = fn(y, 0)
        parts = fn(y[:1], 0) + fn(y[1:], 1)
        same(whole, again)
        same(whole, parts)


def test_torch_rng_mode_consumes_the_generator():
    vox = synth.synthetic_voxels(2, 16, 20, seed=4, regime="stress")
    y = torch.from_numpy(vox).cuda()
    torch.manual_seed(1)
    a = SM.sample_voxel_baseline(y, random=True, rng="torch")
    torch.manual_seed(1)
    b = SM.sample_voxel_baseline(y, random=True, rng="torch")
    c = SM.sample_voxel_baseline(y, random=True, rng="torch")
    same(a, b)
    assert np.asarray(a[0]).tobytes() != np.asarray(c[0]).tobytes()
    # expected counts: floor parts exactly, Bernoulli part within 6 sigma
    n_floor = np.floor(vox).sum(axis=(1, 2, 3, 4))
    frac = vox - np.floor(vox)
    for i in range(2):
        extra = len(a[i]) - n_floor[i]
        mu, sd = frac[i].sum(), np.sqrt((frac[i] * (1 - frac[i])).sum())
        assert abs(extra - mu) < 6 * sd


def test_empty_and_errors():
    y = torch.zeros(2, 2, 10, 8, 9, device="cuda")
    res = SM.sample_voxel_baseline(y, even=True, seed=1)
    assert [len(r) for r in res] == [0, 0] and res[0].dtype.itemsize == 13
    with pytest.raises(AssertionError):
        SM.sample_voxel_baseline(y)
    assert [len(r) for r in SM.sample_voxel_pure_slope(y, pooling_type="avg")] == [0, 0]
    with pytest.raises(hip.V2ceHipError):
        SM.sample_voxel_pure_slope(y, pooling_type="avg", pooling_kernel_size=4)      # even sizes change H x W in the reference
    with pytest.raises(hip.V2ceHipError):
        SM.sample_voxel_baseline(y.cpu(), even=True)
    with pytest.raises(ValueError):
        SM.sample_voxel_baseline(torch.zeros(2, 2, 9, 8, 9, device="cuda"), even=True)
    big = torch.full((1, 2, 10, 4, 4), 3.5, device="cuda")
    with pytest.raises(ValueError):           # u_int too short for floor(y) = 3
        SM.sample_voxel_baseline(big, random=True, u_int=torch.rand(1, 2, 10, 4, 4, 2), u_dec=torch.rand(1, 2, 10, 4, 4),
                                 u_bern=torch.rand(1, 2, 10, 4, 4))


def test_full_size_frame_properties():
    """346x260 stress chunk (the size class of C5): counts = floor parts + Bernoulli hits, every frame sorted
    lexicographically, all coordinates inside the frame."""
    vox = synth.synthetic_voxels(3, 260, 346, seed=8, regime="stress")
    y = torch.from_numpy(vox).cuda()
    ev = SM.sampler_device(y, hip.SAMPLER_PURE_SLOPE, 0, 30, seed=5)
    recs = ev.to_recarrays()
    fold = vox.copy()
    fold[:, :, 8] += fold[:, :, 9]
    fold[:, :, 9] = 0
    n_floor = np.floor(fold).sum(axis=(1, 2, 3, 4))
    for i, r in enumerate(recs):
        assert n_floor[i] <= len(r) <= n_floor[i] + 2 * 10 * 260 * 346
        key = (r["timestamp"].astype(np.int64) << 20) | (r["x"].astype(np.int64) << 10) | r["y"].astype(np.int64)
        key = key * 2 + r["polarity"]
        assert np.all(np.diff(key) >= 0)
        assert r["x"].min() >= 0 and r["x"].max() < 346 and r["y"].min() >= 0 and r["y"].max() < 260
        assert r["timestamp"].min() >= -1 and r["timestamp"].max() < 33400


@pytest.mark.parametrize("name", ["slope_weighted", "slope_avg3", "slope_avg5"])
def test_pooled_pure_slope(gold_dir, name):
    """VERDICT r2 missing #4: pure_slope_sample.py:79-85 on the device (v2ce_sampler_pool + the pooled slope): the
    reference's own events (goldens G10p) with timestamps within 1 us -- the one float path of stage 2 whose last bit
    is the reference's convolution backend's --, and the oracle (same summation order) bit for bit; Philox at a
    larger size as well."""
    z = np.load(os.path.join(gold_dir, f"sampler_g10p_{name}.npz"))
    opts = dict(pooling_type=str(z["pooling_type"]), pooling_kernel_size=int(z["pooling_kernel_size"]))
    draws = dict(u_int=z["u_int"], u_dec=z["u_dec"], u_bern=z["u_bern"])
    res = run_device(z["vox"], "pure_slope", "slope", float(z["t0"]), float(z["fps"]), **draws, **opts)
    assert [len(r) for r in res] == z["lens"].tolist()
    ref = np.frombuffer(z["events"].tobytes(), OS.EVENT_DTYPE)
    lo = 0
    for r in res:
        d = OS.events_close(np.asarray(r), ref[lo:lo + len(r)])
        assert 0 <= d <= len(r) // 1000, d
        lo += len(r)
    same(res, OS.sample_voxel_pure_slope(z["vox"], float(z["t0"]), float(z["fps"]), **draws, **opts))
    vox = synth.synthetic_voxels(2, 33, 47, seed=5, regime="stress")
    M = int(max(np.floor(vox).max(), np.floor(vox[:, :, 8] + vox[:, :, 9]).max()))
    u_int, u_dec, u_bern = OS.philox_draws(2, 33, 47, M, seed=9, frame_base=3)
    want = OS.sample_voxel_pure_slope(vox, 0, 30, u_int=u_int, u_dec=u_dec, u_bern=u_bern, **opts)
    same(run_device(vox, "pure_slope", "slope", 0, 30, seed=9, frame_base=3, **opts), want)
