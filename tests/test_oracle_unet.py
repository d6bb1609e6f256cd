"""CPU: the functional stage-1 oracle (oracle/unet.py) against outputs of the reference V2ce3d
(tests/golden/unet_g1.npz, written by oracle/make_goldens.py)."""
import os

import numpy as np
import torch

from oracle import unet as U
from v2ce_toolbox_amd import synth

TOL = 1e-5   # north_star: "within 1e-5 on float voxel grids" (abs + rel)


def close(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.all(np.abs(a - b) <= TOL + TOL * np.abs(b)), float(np.abs(a - b).max())


def test_state_dict_layout():
    sd = synth.make_state_dict(0)
    assert len(sd) == 218
    assert sum(v.numel() for v in sd.values()) == 52916466
    assert sd["UNet.decoders.0.conv1.module.weight_bar"].shape == (256, 768, 3, 3, 3)
    assert sd["UNet.decoders.0.conv1.module.weight_v"].shape == (20736,)
    assert sd["UNet.encoders.3.downsample.0.weight"].shape == (512, 256, 1, 1, 1)
    assert sd["UNet.encoders.0.bn1.num_batches_tracked"].dtype == torch.int64
    sd2 = synth.make_state_dict(0)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_oracle_matches_reference_three_calls(gold_dir):
    z = np.load(os.path.join(gold_dir, "unet_g1.npz"))
    sd = synth.make_state_dict(0)
    torch.set_num_threads(8)
    out1, inter = U.forward(sd, torch.from_numpy(z["xa"]), return_intermediates=True)
    ok, d = close(out1.numpy(), z["out1"])
    assert ok, d
    for k, v in inter.items():
        ok, d = close(v.numpy(), z["inter_" + k])
        assert ok, (k, d)
    out2 = U.forward(sd, torch.from_numpy(z["xa"]))
    ok, d = close(out2.numpy(), z["out2"])
    assert ok, d
    # spectral-norm state really moved between calls (random-init u/v): SURVEY 8a5
    assert np.abs(z["out2"] - z["out1"]).max() > 1e-4
    out3 = U.forward(sd, torch.from_numpy(z["xb"]))
    ok, d = close(out3.numpy(), z["out3"])
    assert ok, d
    for k in z.files:
        if k.startswith("u_after3_"):
            assert np.allclose(sd[k[len("u_after3_"):]].numpy(), z[k], atol=1e-6)


def test_nearest_index_rule():
    """F.interpolate(nearest) source index == floor(dst*in/out) for the 8 (in,out) pairs used."""
    import torch.nn.functional as F
    for i, o in [(17, 33), (22, 44), (33, 65), (44, 87), (65, 130), (87, 173), (130, 260), (173, 346)]:
        src = F.interpolate(torch.arange(i, dtype=torch.float32).view(1, 1, 1, i), size=(1, o),
                            mode="nearest").view(-1).long().numpy()
        assert np.array_equal(src, (np.arange(o) * i) // o)
