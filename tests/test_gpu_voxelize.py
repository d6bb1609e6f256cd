"""GPU: the HIP voxeliser (v2ce_voxelize_events through the C ABI) against the oracle and the
reference golden; tolerance = f32 summation order (float atomics vs sequential put_)."""
import os

import numpy as np
import pytest
import torch

from oracle import ldati as O
from oracle.voxelize import gen_discretized_event_volume as oracle_vox
from v2ce_toolbox_amd import synth

pytestmark = pytest.mark.gpu


def close(got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(d <= 2e-6 * (1.0 + np.abs(want) * 8)), f"max |d| = {d.max():.3e}"


@pytest.mark.parametrize("name", ["stress", "sparse"])
def test_matches_reference_golden(gold_dir, name):
    from v2ce_toolbox_amd.voxelize import gen_discretized_event_volume
    z = np.load(os.path.join(gold_dir, "voxelize_g8.npz"))
    ev, want = z[f"events_{name}"], z[f"volume_{name}"]
    got = gen_discretized_event_volume(ev, want.shape).cpu().numpy()
    close(got, want)


@pytest.mark.parametrize("bins", [2, 5, 10])
def test_random_events_vs_oracle(bins):
    from v2ce_toolbox_amd.voxelize import gen_discretized_event_volume
    rng = np.random.RandomState(bins)
    n, H, W = 200000, 37, 53
    ev = np.zeros(n, O.EVENT_DTYPE)
    ev["timestamp"] = rng.randint(5, 10 ** 6, n)            # unsorted, beyond 2^16
    ev["timestamp"][:3] = [5, 10 ** 6, 5]                    # exact end points (integer bins)
    ev["x"], ev["y"], ev["polarity"] = rng.randint(0, W, n), rng.randint(0, H, n), rng.randint(0, 2, n)
    got = gen_discretized_event_volume(ev, (2 * bins, H, W)).cpu().numpy()
    close(got, oracle_vox(ev, (2 * bins, H, W)))


def test_errors():
    from v2ce_toolbox_amd.voxelize import gen_discretized_event_volume
    ev = np.zeros(4, O.EVENT_DTYPE)
    ev["timestamp"] = [1, 2, 3, 4]
    ev["x"] = [0, 1, 2, 99]
    with pytest.raises(AssertionError):
        gen_discretized_event_volume(ev, (20, 8, 8))
    ev["x"] = 0
    ev["timestamp"] = 7
    with pytest.raises(RuntimeError):
        gen_discretized_event_volume(ev, (20, 8, 8))
    with pytest.raises(RuntimeError):
        gen_discretized_event_volume(ev[:0], (20, 8, 8))


def test_ldati_round_trip_full_size():
    """Size-independent properties at 346x260 (the authors' own sanity check,
    stage2_metrics.py:187-190): voxelising the LDATI events of one frame-pair gives a volume whose
    total mass is the event count and whose per-(polarity, pixel) mass is that pixel's event count."""
    from v2ce_toolbox_amd.LDATI import ldati_device
    from v2ce_toolbox_amd.voxelize import gen_discretized_event_volume
    vox = torch.from_numpy(synth.synthetic_voxels(1, 260, 346, seed=3, regime="stress")).cuda()
    ev = ldati_device(vox, fps=30, seed=5)
    vol = gen_discretized_event_volume(ev, (20, 260, 346))
    n = ev.num_events
    assert abs(float(vol.double().sum()) - n) < 1e-6 * n
    pix = ev.y.long() * 346 + ev.x.long()
    for pol, planes in ((1, vol[:10]), (0, vol[10:])):
        cnt = torch.bincount(pix[ev.p == pol], minlength=260 * 346).reshape(260, 346).double()
        assert float((planes.double().sum(0) - cnt).abs().max()) < 1e-3
