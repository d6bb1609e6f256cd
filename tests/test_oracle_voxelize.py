"""CPU: the oracle voxeliser against the golden captured from the reference
(train/scripts/utils/events_utils.py:147-175 via oracle/make_goldens.py G8)."""
import os

import numpy as np
import pytest

from oracle import ldati as O
from oracle.voxelize import gen_discretized_event_volume


@pytest.mark.parametrize("name", ["stress", "sparse"])
def test_oracle_matches_reference_golden(gold_dir, name):
    z = np.load(os.path.join(gold_dir, "voxelize_g8.npz"))
    ev = z[f"events_{name}"].view(O.EVENT_DTYPE) if z[f"events_{name}"].dtype == np.uint8 else z[f"events_{name}"]
    want = z[f"volume_{name}"]
    got = gen_discretized_event_volume(ev, want.shape)
    assert got.dtype == np.float32 and np.array_equal(got, want)          # bit-exact, same summation order


def test_mass_and_polarity_planes():
    """Each event contributes weight 1 in total; polarity 0 lands in the second half of the volume."""
    rng = np.random.RandomState(0)
    n = 5000
    ev = np.zeros(n, O.EVENT_DTYPE)
    ev["timestamp"] = np.sort(rng.randint(100, 33000, n))
    ev["x"], ev["y"], ev["polarity"] = rng.randint(0, 14, n), rng.randint(0, 12, n), rng.randint(0, 2, n)
    vol = gen_discretized_event_volume(ev, (20, 12, 14))
    assert abs(vol.sum() - n) < 1e-2
    assert abs(vol[:10].sum() - (ev["polarity"] == 1).sum()) < 1e-2
    assert abs(vol[10:].sum() - (ev["polarity"] == 0).sum()) < 1e-2
