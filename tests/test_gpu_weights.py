"""GPU parity at FULL size (one 346x260 sequence, T = 16, the default split-half path) on weight states other than
``synth.make_state_dict(0)`` -- VERDICT r4 #6.  The pretrained ``weights/v2ce_3d.pt`` is a download that is absent from the
reference checkout (/root/reference/readme.md:13), so no test can use it; what can be had are the distribution families a
trained checkpoint may come from:

* other seeds of the Gaussian generator;
* a heavy-tailed state: Student-t_4 weights, a tenth of the BatchNorm channels with running variances down to 1e-3 and
  folded scales spread over a factor of four, ``pred`` gain 4 (voxel counts up to ~9) -- it must stay under the range guard;
* a state that trips the guard: the call is repeated on the exact-f32 kernels (equal to ``precision='f32'`` bit for bit),
  its cost is logged;
* spectral-norm u / v converged by 50 power iterations on weights with a dominant singular direction (where a trained
  checkpoint's are; the generator's matrices are i.i.d. Gaussian -- nearly degenerate leading singular values -- and its u / v
  random unit vectors): the oracle at 1e-5, and ``dist.fast_forward`` -- skipping the convolutions of k calls -- is a no-op inside 1e-5.

Reference forward: /root/reference/scripts/unet_2layer.py:335-379, scripts/spectral_norm.py:19-31, v2ce.py:30-43.
Bar: 1e-5 abs + 1e-5 rel against oracle/unet.py (torch CPU f32), like tests/test_gpu_fullsize.py."""
import logging
import os
import time

import numpy as np
import pytest
import torch

from oracle import glue as OG
from oracle import unet as U
from v2ce_toolbox_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5
H, W = 260, 346


def excess(a, b, tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((np.abs(a - b) - tol * np.abs(b)).max())


def load(sd, precision="f16x2", guard="call"):
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d(precision=precision, guard=guard)
    m.load_state_dict(sd, strict=True)
    return m.eval().to("cuda")


@pytest.fixture(scope="module")
def x_full():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    return OG.preprocess(synth.synthetic_frames(17, H, W, seed=77))[None]          # [1,16,2,H,W]


def clone(sd):
    return {k: v.clone() for k, v in sd.items()}


STATES = {
    "seed1": lambda: synth.make_state_dict(1),
    "seed2": lambda: synth.make_state_dict(2),
    "student_t4_gain4": lambda: synth.make_state_dict(3, out_gain=4.0, tails="student"),
    "converged_uv": lambda: synth.converge_spectral_norm(synth.make_state_dict(0), 50, separate=2.0),
}


@pytest.mark.parametrize("name", list(STATES))
def test_full_size_default_path_vs_oracle(name, x_full):
    sd = STATES[name]()
    want = U.forward(clone(sd), torch.from_numpy(x_full)).contiguous().numpy()
    m = load(sd)
    got = m(torch.from_numpy(x_full).cuda()).cpu().numpy()
    assert got.shape == (1, 16, 20, H, W)
    assert m.guard_reruns == 0, "this state must stay under the range guard"
    bound = m.range_guard_value()
    print(f"{name}: max voxel {want.max():.2f}, share > 1: {(want > 1).mean():.4f}, guard bound {bound:.2e}, "
          f"max |d| {np.abs(got - want).max():.2e}")
    assert 0 < bound <= m.RANGE_GUARD_LIMIT
    assert want.max() > 1.0                                        # multi-event voxels exist
    assert excess(got, want) <= TOL, excess(got, want)


def test_full_size_guard_trip_reruns_on_exact_f32(x_full, caplog):
    """A heavy-tailed state pushed over the guard's limit (one encoder channel at 3e6 next to O(1) ones): the bare call
    notices, rewinds the spectral-norm state and repeats itself on the exact-f32 kernels."""
    sd = synth.make_state_dict(3, out_gain=4.0, tails="student")
    sd["UNet.encoders.0.downsample.0.bias"][3] = 3e6
    x = torch.from_numpy(x_full).cuda()
    mg, mx = load(sd, "f16x2"), load(sd, "f32")
    t0 = time.perf_counter()
    want = mx(x)
    torch.cuda.synchronize()
    t_exact = time.perf_counter() - t0
    with caplog.at_level(logging.WARNING, logger="V2CE"):
        t0 = time.perf_counter()
        got = mg(x)
        torch.cuda.synchronize()
        t_guarded = time.perf_counter() - t0
    assert mg.guard_reruns == 1 and any("range guard" in r.message for r in caplog.records)
    assert torch.equal(got, want)
    for (n, p), (_, q) in zip(mg.named_parameters(), mx.named_parameters()):
        if n.endswith(("weight_u", "weight_v")):
            assert torch.equal(p, q), n
    # (against the oracle this state only admits a bar relative to the tensor's range: the 3e6 channel makes the voxels
    # differences of numbers of 1e5, in the reference's own f32 arithmetic as much as here)
    ref = U.forward(clone(sd), torch.from_numpy(x_full)).contiguous().numpy()
    assert np.abs(got.cpu().numpy() - ref).max() <= TOL * np.abs(ref).max()
    print(f"guard trip at full size: split-half attempt + exact-f32 rerun {1e3 * t_guarded:.1f} ms (first calls: includes the "
          f"derived-constant rebuild) vs exact f32 alone {1e3 * t_exact:.1f} ms")


def test_converged_uv_make_fast_forward_a_no_op(x_full):
    """With u / v at the fixed point of the power iteration a further iteration moves sigma by rounding only: a replica that
    skips the convolutions of three calls (dist.fast_forward -> advance_spectral_norm) and one that runs them give the same
    voxels inside the parity bar -- the regime of a trained checkpoint, where sharding calls over GPUs cannot show."""
    from v2ce_toolbox_amd import dist as vdist
    sd = synth.converge_spectral_norm(synth.make_state_dict(0), 50, separate=2.0)
    x = torch.from_numpy(x_full).cuda()
    a, b, c = load(sd, guard="deferred"), load(sd, guard="deferred"), load(sd, guard="deferred")
    first = a(x)
    for _ in range(3):
        b(x)                                                        # three real calls ...
    vdist.fast_forward(c, 3)                                        # ... or three power iterations without the convolutions
    assert b.calls == c.calls == 3
    yb, yc = b(x), c(x)
    assert torch.equal(yb, yc)                                      # the trajectory does not depend on the input
    d = (yb - first).abs()
    assert float((d - TOL * first.abs()).max()) <= TOL, float(d.max())      # call 4 == call 1 inside the bar: converged
    # and the generator's own (random) u / v are NOT converged: there the call index matters (what fast_forward is for)
    r = load(synth.make_state_dict(0), guard="deferred")
    y1 = r(x)
    r(x), r(x)
    y4 = r(x)
    assert float((y4 - y1).abs().max()) > 1e-3
