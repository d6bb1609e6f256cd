"""GPU parity: HIP stage 1 (conv3d / spectral norm / V2ce3d, through the C ABI) vs the torch-fp32
oracle and the goldens captured from the reference.  Tolerance (north_star): 1e-5 abs + 1e-5 rel."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import glue as OG
from oracle import unet as U
from v2ce_toolbox_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def assert_close(a, b, what="", tol=TOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b) - tol * np.abs(b)
    i = np.unravel_index(np.argmax(err), err.shape)
    assert err[i] <= tol, f"{what}: max excess at {i}: got {a[i]!r} want {b[i]!r} (|d|={abs(a[i]-b[i]):.3e})"


def to_btchw(x_ncdhw):
    return x_ncdhw.permute(0, 2, 1, 3, 4).contiguous()


def hip_conv(x0, w, scale, shift, ksize, stride, act, x1=None, up_to=None, residual=None):
    """x0/x1/residual given as NCDHW torch CPU tensors; returns NCDHW numpy."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps = {}
    m.precision = "f32"
    wd = w.cuda().contiguous()
    wp = V2ce3d._pack(m, wd)
    y = V2ce3d._conv(m, to_btchw(x0).cuda(), None if x1 is None else to_btchw(x1).cuda(), wp,
                     scale.cuda().contiguous(), shift.cuda().contiguous(), w.shape[0], ksize, stride,
                     act, residual=None if residual is None else to_btchw(residual).cuda(), up_to=up_to, dense_out=True)
    torch.cuda.synchronize()
    return y.permute(0, 2, 1, 3, 4).cpu().numpy()


def ref_conv(x0, w, scale, shift, ksize, stride, act, x1=None, up_to=None, residual=None):
    x = x0.double()
    if up_to is not None:
        x = U.upsample_nearest_hw(x0, up_to).double()
    if x1 is not None:
        x = torch.cat([x, x1.double()], dim=1)
    y = F.conv3d(x, w.double(), None, (1, stride, stride), ksize // 2)
    y = y * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if residual is not None:
        y = y + residual.double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, 0.01)
    return y.numpy()


CONV_CASES = [
    # B, T, Cin, Cout, H, W, k, s, act, residual
    (1, 4, 2, 32, 20, 28, 3, 1, 2, False),      # head-like (Cin=2): the dedicated head kernel
    (2, 5, 2, 32, 37, 71, 3, 1, 2, False),      # head kernel, ragged in T, H and W (4 x 4 x 32 boxes), batch 2
    (1, 3, 2, 32, 9, 33, 3, 1, 1, True),        # head kernel with a residual and ReLU
    (2, 3, 32, 64, 19, 23, 3, 2, 1, False),     # encoder conv1 (stride 2, odd sizes)
    (1, 16, 64, 64, 9, 11, 3, 1, 1, True),      # conv2 + residual + relu
    (1, 2, 32, 64, 19, 23, 1, 2, 0, False),     # strided 1x1 shortcut
    (1, 5, 32, 20, 33, 47, 1, 1, 1, False),     # pred-like (Cout=20)
    (1, 2, 128, 128, 17, 22, 3, 1, 0, False),   # bottleneck-like spatial size
    (1, 16, 32, 32, 40, 70, 3, 1, 1, True),     # Cout=32 full-depth tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_vs_f64(case):
    B, T, Cin, Cout, H, W, k, s, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.2 * torch.randn(Cout, generator=g)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = torch.randn(B, Cout, T, Ho, Wo, generator=g) if use_res else None
    got = hip_conv(x, w, scale, shift, k, s, act, residual=res)
    want = ref_conv(x, w, scale, shift, k, s, act, residual=res)
    assert_close(got, want, str(case))


@pytest.mark.parametrize("k", [1, 3])
def test_conv3d_virtual_upsample_concat(k):
    """Decoder input: nearest-upsample(x0) ++ skip, never materialised (unet_2layer.py:358-364)."""
    g = torch.Generator().manual_seed(5 + k)
    x0 = torch.randn(1, 64, 3, 17, 22, generator=g)
    x1 = torch.randn(1, 32, 3, 33, 44, generator=g)
    w = torch.randn(32, 96, k, k, k, generator=g) * (2.0 / (96 * k ** 3)) ** 0.5
    scale, shift = torch.ones(32), torch.zeros(32)
    got = hip_conv(x0, w, scale, shift, k, 1, 1, x1=x1, up_to=(33, 44))
    want = ref_conv(x0, w, scale, shift, k, 1, 1, x1=x1, up_to=(33, 44))
    assert_close(got, want, f"upsample+concat k={k}")


def test_sn_power_iteration_and_pack():
    from v2ce_toolbox_amd import hip
    g = torch.Generator().manual_seed(3)
    rows, cols = 64, 96 * 27
    w = torch.randn(rows, 96, 3, 3, 3, generator=g) * 0.05
    u = torch.randn(rows, generator=g); u /= u.norm()
    v = torch.randn(cols, generator=g); v /= v.norm()
    sd = {"m.weight_u": u.clone(), "m.weight_v": v.clone(), "m.weight_bar": w.clone()}
    L = hip.lib()
    ud, vd, wd = u.cuda(), v.cuda(), w.cuda().contiguous()
    ws = torch.empty(L.v2ce_sn_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")
    sig = torch.empty(1, device="cuda")
    wp = torch.empty(rows * cols, device="cuda")
    for it in range(3):
        wn = U.sn_step(sd, "m")
        hip.check(L.v2ce_sn_power_iter(ud.data_ptr(), vd.data_ptr(), wd.data_ptr(), rows, cols,
                                       sig.data_ptr(), ws.data_ptr(), ws.numel(), hip.stream_ptr()), "sn")
        hip.check(L.v2ce_pack_weights(wd.data_ptr(), rows, 96, 27, sig.data_ptr(), wp.data_ptr(),
                                      hip.stream_ptr()), "pack")
        torch.cuda.synchronize()
        assert_close(ud.cpu().numpy(), sd["m.weight_u"].numpy(), f"u it{it}", 2e-6)
        assert_close(vd.cpu().numpy(), sd["m.weight_v"].numpy(), f"v it{it}", 2e-6)
        want = wn.reshape(rows, 96, 27).permute(1, 2, 0).contiguous().numpy()      # [ci][tap][co]
        assert_close(wp.cpu().numpy().reshape(96, 27, rows), want, f"w/sigma it{it}", 2e-6)


def load_model(precision="f32"):
    """The exact-f32 arithmetic path (the split-half default has its own tests below)."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d(precision=precision)
    m.load_state_dict(synth.make_state_dict(0), strict=True)
    return m.eval().to("cuda")


def test_v2ce3d_matches_reference_three_calls(gold_dir):
    """G1: outputs of the reference V2ce3d for three consecutive calls (SN state advances)."""
    z = np.load(os.path.join(gold_dir, "unet_g1.npz"))
    m = load_model()
    out1, inter = m(torch.from_numpy(z["xa"]).cuda(), return_intermediates=True)
    torch.cuda.synchronize()
    for k, v in inter.items():
        assert_close(v.permute(0, 2, 1, 3, 4).cpu().numpy(), z["inter_" + k], k)
    assert_close(out1.cpu().numpy(), z["out1"], "call 1")
    out2 = m(torch.from_numpy(z["xa"]).cuda())
    assert_close(out2.cpu().numpy(), z["out2"], "call 2")
    out3 = m(torch.from_numpy(z["xb"]).cuda())
    assert_close(out3.cpu().numpy(), z["out3"], "call 3")
    assert m.calls == 3
    sd = m.state_dict()
    for k in z.files:
        if k.startswith("u_after3_"):
            assert_close(sd[k[len("u_after3_"):]].cpu().numpy(), z[k], k, 2e-6)


def test_v2ce3d_state_dict_roundtrip():
    m = load_model()
    x = torch.randn(1, 2, 2, 16, 16, device="cuda")
    m(x)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m2 = V2ce3d(precision="f32")
    m2.load_state_dict(sd)
    m2 = m2.eval().to("cuda")
    assert torch.equal(m(x), m2(x))


def test_v2ce3d_full_width_tile_vs_oracle():
    """One 346x260 sequence slice (L=2 to keep the CPU oracle fast) against oracle/unet.py."""
    sd = synth.make_state_dict(0)
    fr = synth.synthetic_frames(3, 260, 346, seed=1)
    x = fr.astype(np.float32) / 255
    x = np.stack([x[:-1], x[1:]], axis=1)
    x = ((x - np.float32(0.153)) / np.float32(0.165))[None]
    want = U.forward(U.clone_state(sd), torch.from_numpy(x)).numpy()
    m = load_model()
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert (want > 1).mean() > 1e-3          # the fixture exercises multi-event voxels
    assert_close(got, want, "346x260 L=2")


# ---- opt-in split-half precision (V2CE_PRECISION_F16X2): same 1e-5 bar -------------------------
def hip_conv_split(x0, w, scale, shift, stride, act, x1=None, up_to=None, residual=None, tracked=False, ksize=3):
    """tracked: pass max|x| slots like V2ce3d does (dynamic power-of-two pre-scale); otherwise the
    kernel's fixed pre-scale (|x| < 4094)."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps = {}
    m.precision = "f16x2" if tracked else "f32"
    m._prep = {"absmax": torch.zeros((4, 2), device="cuda")}
    m._slot = 0
    wq = V2ce3d._pack(m, w.cuda().contiguous(), split=True)
    x0d = to_btchw(x0).cuda()
    x1d = None if x1 is None else to_btchw(x1).cuda()
    if tracked:
        x0d.absmax = x0d.abs().max().reshape(1)
        if x1d is not None:
            x1d.absmax = x1d.abs().max().reshape(1)
    # the split-half kernels take and produce the channels-last-16 layout
    c16 = V2ce3d.to_c16
    y = V2ce3d._conv(m, c16(x0d), None if x1d is None else c16(x1d), wq,
                     scale.cuda().contiguous(), shift.cuda().contiguous(), w.shape[0], ksize, stride, act,
                     residual=None if residual is None else c16(to_btchw(residual).cuda()), up_to=up_to,
                     split=True, dense_out=True)
    torch.cuda.synchronize()
    hip_conv_split.guard = float(y.absmax[1])               # the launch's range-guard bound
    return V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()


SPLIT_CASES = [
    # B, T, Cin, Cout, H, W, s, act, residual
    (1, 16, 64, 64, 9, 11, 1, 1, True),
    (2, 3, 32, 64, 19, 23, 2, 1, False),
    (1, 2, 128, 128, 17, 22, 1, 0, False),
    (1, 16, 32, 32, 40, 70, 1, 1, True),
    (1, 4, 64, 32, 21, 30, 2, 0, False),
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv3d_split_half_vs_f64(case):
    B, T, Cin, Cout, H, W, s, act, use_res = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.2 * torch.randn(Cout, generator=g)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Cout, T, Ho, Wo, generator=g) if use_res else None
    got = hip_conv_split(x, w, scale, shift, s, act, residual=res)
    want = ref_conv(x, w, scale, shift, 3, s, act, residual=res)
    assert_close(got, want, "split " + str(case))


def test_conv3d_split_half_virtual_concat():
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(1, 64, 3, 17, 22, generator=g)
    x1 = torch.randn(1, 32, 3, 33, 44, generator=g)
    w = torch.randn(32, 96, 3, 3, 3, generator=g) * (2.0 / (96 * 27)) ** 0.5
    got = hip_conv_split(x0, w, torch.ones(32), torch.zeros(32), 1, 1, x1=x1, up_to=(33, 44))
    want = ref_conv(x0, w, torch.ones(32), torch.zeros(32), 3, 1, 1, x1=x1, up_to=(33, 44))
    assert_close(got, want, "split upsample+concat")


@pytest.mark.parametrize("case", [
    # B, T, C0, C1, Cout, H, W, stride, upsampled source 0
    (1, 5, 32, 0, 64, 21, 30, 2, False),     # encoder shortcut: strided gather
    (2, 3, 64, 32, 32, 19, 26, 1, True),     # decoder shortcut: virtual upsample + concat, 32 channels
    (1, 16, 128, 0, 128, 9, 11, 1, False),
    (1, 4, 64, 0, 128, 18, 23, 2, False),
])
def test_conv1x1_split_half_vs_f64(case):
    """The 1x1x1 shortcut convs on the split-half kernel (one tap per chunk, strided / virtual gathers)."""
    B, T, C0, C1, Cout, H, W, s, ups = case
    g = torch.Generator().manual_seed(11)
    h0, w0 = ((H + 1) // 2, (W + 1) // 2) if ups else (H, W)
    x0 = torch.randn(B, C0, T, h0, w0, generator=g)
    x1 = torch.randn(B, C1, T, H, W, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, 1, 1, 1, generator=g) * (2.0 / (C0 + C1)) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.2 * torch.randn(Cout, generator=g)
    up_to = (H, W) if ups else None
    got = hip_conv_split(x0, w, scale, shift, s, 0, x1=x1, up_to=up_to, tracked=True, ksize=1)
    want = ref_conv(x0, w, scale, shift, 1, s, 0, x1=x1, up_to=up_to)
    assert_close(got, want, "split 1x1 " + str(case))


def _sweep_cases():
    rng = np.random.RandomState(2024)
    cases = []
    for _ in range(14):
        ks = int(rng.choice([1, 3]))
        s = int(rng.choice([1, 2]))
        ups = bool(rng.rand() < 0.35) and s == 1
        c0 = int(rng.choice([16, 32, 48, 64, 96, 128]))
        c1 = int(rng.choice([16, 32, 64])) if ups else 0
        cout = int(rng.choice([32, 64, 96, 128, 160]))
        cases.append((int(rng.randint(1, 3)), int(rng.randint(1, 6)), c0, c1, cout, int(rng.randint(5, 40)),
                      int(rng.randint(5, 50)), ks, s, ups, bool(rng.rand() < 0.5) and ks == 3 and s == 1))
    return cases


@pytest.mark.parametrize("case", _sweep_cases())
def test_split_half_shape_sweep(case):
    """Randomised shapes for the persistent wave-specialised kernel: odd planes, T < tile, several
    (also non-power-of-two) channel tiles, partial last channel tile, strided / virtual gathers."""
    B, T, C0, C1, Cout, H, W, ks, s, ups, use_res = case
    g = torch.Generator().manual_seed(abs(hash(case)) % 100000)
    h0, w0 = ((H + 1) // 2, (W + 1) // 2) if ups else (H, W)
    x0 = torch.randn(B, C0, T, h0, w0, generator=g)
    x1 = torch.randn(B, C1, T, H, W, generator=g) if C1 else None
    w = torch.randn(Cout, C0 + C1, ks, ks, ks, generator=g) * (2.0 / ((C0 + C1) * ks ** 3)) ** 0.5
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = 0.2 * torch.randn(Cout, generator=g)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Cout, T, Ho, Wo, generator=g) if use_res else None
    up_to = (H, W) if ups else None
    got = hip_conv_split(x0, w, scale, shift, s, 1, x1=x1, up_to=up_to, residual=res, tracked=True, ksize=ks)
    want = ref_conv(x0, w, scale, shift, ks, s, 1, x1=x1, up_to=up_to, residual=res)
    assert_close(got, want, "sweep " + str(case))


@pytest.mark.parametrize("mag", [1e-6, 1.0, 3e4, 1e9])
def test_conv3d_split_half_dynamic_range(mag):
    """Activations and weights far outside the fp16 range: the tracked power-of-two pre-scales keep
    the split-half result at f32 accuracy (relative to the output magnitude) and the launch records
    max |y| for its consumer."""
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(1, 32, 2, 9, 12, generator=g) * mag
    x1 = torch.randn(1, 32, 2, 17, 23, generator=g) * (mag * 1e-3)
    w = torch.randn(64, 64, 3, 3, 3, generator=g) * (300.0 / mag)      # |w| beyond 256 for small mag
    one, zero = torch.ones(64), torch.zeros(64)
    got = hip_conv_split(x0, w, one, zero, 1, 0, x1=x1, up_to=(17, 23), tracked=True)
    want = ref_conv(x0, w, one, zero, 3, 1, 0, x1=x1, up_to=(17, 23))
    assert np.isfinite(got).all()
    ref_mag = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-6 * ref_mag


@pytest.mark.parametrize("outlier", [1e4, 1e6, 1e8])
def test_conv3d_split_half_single_outlier(outlier):
    """ONE huge activation among O(1) values: the tensor-wide power-of-two
    pre-scale is set by the outlier, so the O(1) values sit `r = log2(outlier)` binades below the
    maximum; their lo half goes fp16-subnormal beyond r = 17 and keeps 22 - (r - 17) bits.  Outputs
    whose receptive field misses the outlier are checked at the north_star bar (1e-5 abs + 1e-5 rel);
    the print shows the measured error for the record (documented bound: include/v2ce_hip.h)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 32, 4, 24, 30, generator=g)
    x[0, 5, 2, 12, 15] = outlier
    w = torch.randn(64, 32, 3, 3, 3, generator=g) * (2.0 / (32 * 27)) ** 0.5
    one, zero = torch.ones(64), torch.zeros(64)
    got = hip_conv_split(x, w, one, zero, 1, 0, tracked=True)
    want = ref_conv(x, w, one, zero, 3, 1, 0)
    far = np.ones(want.shape, bool)
    far[:, :, 1:4, 11:14, 14:17] = False                       # outputs that see the outlier
    err = np.abs(got - want)
    excess_far = (err - TOL * np.abs(want))[far].max()
    rel_near = (err / np.abs(want).clip(1e-30))[~far].max()
    print(f"outlier {outlier:g}: max |err| away from it {err[far].max():.3e}, max rel err at it {rel_near:.3e}")
    assert rel_near <= 1e-5
    # the guard bound is a worst case: it already trips at 1e4 (measured error still < 1e-6), so every
    # case whose error can exceed the bar is routed to the exact-f32 kernels (test_range_guard_*)
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    print(f"guard bound {hip_conv_split.guard:.3e}")
    assert hip_conv_split.guard > V2ce3d.RANGE_GUARD_LIMIT
    assert hip_conv_split.guard >= err[far].max()
    if outlier <= 1e6:
        assert excess_far <= TOL, excess_far
    else:          # 27 binades: ~12 bits left for the O(1) values -- the documented limit of one scale per tensor
        assert excess_far <= 5e-3


def test_conv3d_split_half_heavy_tail():
    """Heavy-tailed activations (Student-t, 2 degrees of freedom: max/median ~ 1e3..1e4) stay at the bar."""
    g = torch.Generator().manual_seed(12)
    z = torch.randn(1, 64, 3, 20, 26, generator=g)
    chi = (torch.randn(z.shape, generator=g) ** 2 + torch.randn(z.shape, generator=g) ** 2) / 2
    x = z / chi.sqrt().clamp_min(1e-3)
    w = torch.randn(64, 64, 3, 3, 3, generator=g) * (2.0 / (64 * 27)) ** 0.5
    one, zero = torch.ones(64), torch.zeros(64)
    got = hip_conv_split(x, w, one, zero, 1, 0, tracked=True)
    want = ref_conv(x, w, one, zero, 3, 1, 0)
    assert float(x.abs().max() / x.abs().median()) > 300
    assert_close(got, want, "heavy tail")


def test_range_guard_quiet_on_ordinary_activations():
    """N(0,1) activations, He-scaled weights: the bound stays two orders of magnitude under the limit and
    above the measured error."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    g = torch.Generator().manual_seed(13)
    for cin, cout in ((32, 64), (256, 128)):
        x = torch.randn(1, cin, 3, 12, 14, generator=g)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (cin * 27)) ** 0.5
        one, zero = torch.ones(cout), torch.zeros(cout)
        got = hip_conv_split(x, w, one, zero, 1, 0, tracked=True)
        assert 0 < hip_conv_split.guard < V2ce3d.RANGE_GUARD_LIMIT / 10, hip_conv_split.guard
        assert_close(got, ref_conv(x, w, one, zero, 3, 1, 0), "guard quiet")


def test_range_guard_reroutes_adversarial_model_to_exact_f32(caplog):
    """a checkpoint whose first encoder block emits one channel at 3e6 next to O(1)
    channels (a 22-binade range inside one tensor).  The split-half launches report it, and the guarded
    entry point (glue.video_to_voxels -> run_guarded) returns what the exact-f32 model returns, bit for
    bit, with the spectral-norm trajectory of a single pass; ordinary weights do not trip."""
    import logging
    from v2ce_toolbox_amd import glue
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    frames = synth.synthetic_frames(18, 32, 48, seed=4)
    sd = synth.make_state_dict(0)
    bad = {k: v.clone() for k, v in sd.items()}
    bad["UNet.encoders.0.downsample.0.bias"][3] = 3e6

    def model(state, precision):
        m = V2ce3d(precision=precision)
        m.load_state_dict(state, strict=True)
        return m.eval().to("cuda")

    m = model(sd, "f16x2")
    glue.video_to_voxels(m, frames=frames, width=48, height=32, batch_size=2)
    quiet = m.range_guard_value()
    assert quiet == 0.0                                   # run_guarded consumed it ...
    m(torch.from_numpy(OG.preprocess(frames[:17])[None]).cuda())
    ordinary = m.range_guard_value()
    print(f"ordinary weights: guard bound {ordinary:.3e}")
    assert 0 < ordinary < V2ce3d.RANGE_GUARD_LIMIT         # ... and a plain forward shows the ordinary level

    mg, mx = model(bad, "f16x2"), model(bad, "f32")
    with caplog.at_level(logging.WARNING, logger="V2CE"):
        got = glue.video_to_voxels(mg, frames=frames, width=48, height=32, batch_size=2)
    assert any("range guard" in r.message for r in caplog.records)
    want = glue.video_to_voxels(mx, frames=frames, width=48, height=32, batch_size=2)
    assert torch.equal(got, want)
    assert mg.calls == mx.calls and mg.precision == "f16x2"
    for (n, p), (_, q) in zip(mg.named_parameters(), mx.named_parameters()):
        if n.endswith(("weight_u", "weight_v")):
            assert torch.equal(p, q), n
    # without the guard the split-half result is outside the bar here: the reroute is needed
    unguarded = V2ce3d(precision="f16x2", guard="deferred")
    unguarded.load_state_dict(bad, strict=True)
    raw = unguarded.eval().to("cuda")(torch.from_numpy(OG.preprocess(frames[:17])[None]).cuda())
    ex = model(bad, "f32")(torch.from_numpy(OG.preprocess(frames[:17])[None]).cuda())
    d = (raw - ex).abs()
    print(f"unguarded split-half vs exact on the adversarial checkpoint: max abs {float(d.max()):.3e}, "
          f"max rel {float((d / ex.abs().clamp_min(1e-30)).max()):.3e}")


def test_bare_model_call_keeps_the_f32_contract(caplog):
    """a REFERENCE-STYLE caller -- ``model = V2ce3d(); model(x)``, then
    ``sample_voxel_statistical`` (v2ce.py:81-82,353-357), no glue.run_guarded anywhere -- on the adversarial
    checkpoint of the test above: the default per-call guard repeats the offending call on the exact-f32 kernels,
    so voxels, events and the spectral-norm trajectory equal those of precision='f32'; call after call."""
    import logging
    from v2ce_toolbox_amd.LDATI import sample_voxel_statistical
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    frames = synth.synthetic_frames(18, 32, 48, seed=4)
    bad = {k: v.clone() for k, v in synth.make_state_dict(0).items()}
    bad["UNet.encoders.0.downsample.0.bias"][3] = 3e6

    def model(**kw):
        m = V2ce3d(**kw)
        m.load_state_dict(bad, strict=True)
        return m.eval().to("cuda")

    mg, mx = model(), model(precision="f32")                # the drop-in default vs exact f32
    assert mg.precision == "f16x2" and mg.guard == "call"
    for k in range(2):                                      # two consecutive calls: the SN state must track as well
        x = torch.from_numpy(OG.preprocess(frames[k:k + 17])[None]).cuda()
        with caplog.at_level(logging.WARNING, logger="V2CE"):
            got = mg(x)
        want = mx(x)
        assert torch.equal(got, want)
        # (the adversarial checkpoint's voxels are huge: clamp them, or LDATI is asked for billions of events)
        ev_g = sample_voxel_statistical(got.reshape(-1, 2, 10, 32, 48).clamp(max=3.0), fps=30, seed=5)
        ev_w = sample_voxel_statistical(want.reshape(-1, 2, 10, 32, 48).clamp(max=3.0), fps=30, seed=5)
        assert all(a.tobytes() == b.tobytes() for a, b in zip(ev_g, ev_w))
    assert mg.guard_reruns == 2 and mg.calls == mx.calls == 2 and mg.precision == "f16x2"
    assert any("range guard" in r.message for r in caplog.records)
    for (n, p), (_, q) in zip(mg.named_parameters(), mx.named_parameters()):
        if n.endswith(("weight_u", "weight_v")):
            assert torch.equal(p, q), n
    # ordinary weights: no rerun, and the guarded call returns what the unguarded one does
    sd = synth.make_state_dict(0)
    a, b = V2ce3d(), V2ce3d(guard="deferred")
    a.load_state_dict(sd), b.load_state_dict(sd)
    x = torch.from_numpy(OG.preprocess(frames[:17])[None]).cuda()
    assert torch.equal(a.eval().to("cuda")(x), b.eval().to("cuda")(x)) and a.guard_reruns == 0


def test_forward_is_batch_invariant():
    """One range slot per batch element (desc.absmax_batch_stride): sequence b of a batch of four gets, bit for
    bit, the voxels it gets alone -- what lets pipeline.run_clip share a reference batch out over GPUs sequence by
    sequence.  Both precisions; the sequences differ in range by 2^6 so that a shared scale would show."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    xs = np.stack([OG.preprocess(synth.synthetic_frames(17, 40, 56, seed=40 + s)) for s in range(4)])
    xs[1] *= 64.0
    xs[3] *= 1.0 / 64.0
    x = torch.from_numpy(xs).cuda()
    for precision in ("f16x2", "f32"):
        def fresh():
            m = V2ce3d(precision=precision)
            m.load_state_dict(synth.make_state_dict(0))
            return m.eval().to("cuda")
        whole = fresh()(x)
        for b in range(4):
            alone = fresh()(x[b:b + 1].contiguous())
            assert torch.equal(whole[b], alone[0]), (precision, b)
        pair = fresh()(x[1:3].contiguous())
        assert torch.equal(whole[1:3], pair), precision


def test_conv3d_records_output_absmax():
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 16, 3, 10, 13, generator=g)
    w = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.1
    for split in (False, True):
        m = V2ce3d.__new__(V2ce3d)
        torch.nn.Module.__init__(m)
        m._maps, m.precision, m._slot = {}, "f16x2", 0
        m._prep = {"absmax": torch.zeros((4, 2), device="cuda")}
        xd = to_btchw(x).cuda()
        xd.absmax = xd.abs().max().reshape(1)
        wp = V2ce3d._pack(m, w.cuda().contiguous(), split=split)
        y = V2ce3d._conv(m, V2ce3d.to_c16(xd) if split else xd, None, wp, torch.ones(32).cuda(), torch.zeros(32).cuda(), 32, 3, 1, 2,
                         split=split, track=True, dense_out=True)
        assert float(y.absmax[0]) == float(y.abs().max())


def test_v2ce3d_split_half_matches_reference_three_calls(gold_dir):
    """The reference goldens at the same 1e-5 bar with precision='f16x2'."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    z = np.load(os.path.join(gold_dir, "unet_g1.npz"))
    m = V2ce3d(precision="f16x2")
    m.load_state_dict(synth.make_state_dict(0), strict=True)
    m = m.eval().to("cuda")
    for x, want in ((z["xa"], z["out1"]), (z["xa"], z["out2"]), (z["xb"], z["out3"])):
        got = m(torch.from_numpy(x).cuda()).cpu().numpy()
        assert_close(got, want, "split-half golden")


_SWITCH_SCRIPT = """
import sys, numpy as np, torch
sys.path.insert(0, '.')
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd.v2ce_3d import V2ce3d
m = V2ce3d(); m.load_state_dict(synth.make_state_dict(0), strict=True); m = m.eval().to('cuda')
x = torch.randn(1, 16, 2, 70, 90, generator=torch.Generator().manual_seed(5)).cuda()
np.save(sys.argv[1], m(x).cpu().numpy())
"""


@pytest.mark.parametrize("switch", ["V2CE_PEPI=0", "V2CE_PEPI_SC=1", "V2CE_G4=1"])
def test_shared_epilogue_switches_agree(switch, tmp_path):
    """Round 6: the conv with the fused head shares its epilogue between consumer and producer waves (default; V2CE_PEPI=0 = the
    consumers' own), the strided convs with the fused shortcut can (V2CE_PEPI_SC=1, measured slower); V2CE_G4=1 = four lanes per
    element in the producers' gather (measured no faster).  The switches are read
    once per process, so each side runs in its own interpreter; the lean epilogue rounds differently (a pre-scale per position
    instead of per wave, contracted multiply-adds): equal to f32 rounding."""
    import subprocess
    import sys
    outs = []
    for env_kv in (None, switch):
        env = dict(os.environ)
        env.pop("V2CE_PEPI", None)
        env.pop("V2CE_PEPI_SC", None)
        env.pop("V2CE_G4", None)
        if env_kv:
            k, v = env_kv.split("=")
            env[k] = v
        f = str(tmp_path / f"out_{len(outs)}.npy")
        subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT, f], check=True, env=env, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        outs.append(np.load(f))
    assert outs[0].shape == (1, 16, 20, 70, 90)
    assert_close(outs[0], outs[1], switch, 2e-6)


def test_fused_head_matches_unfused():
    """Default path: `pred` fused into the last decoder conv (v2ce_conv3d_fwd_pred).  Asking for the
    intermediates runs the layers separately; both must agree to f32 rounding."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    x = torch.randn(2, 5, 2, 37, 50, generator=torch.Generator().manual_seed(2)).cuda()
    outs = []
    for inter in (False, True):
        m = V2ce3d()
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        m = m.eval().to("cuda")
        assert m.precision == "f16x2"
        o = m(x, return_intermediates=inter)
        outs.append((o[0] if inter else o).cpu().numpy())
    assert outs[0].shape == (2, 5, 20, 37, 50)
    assert_close(outs[0], outs[1], "fused vs separate head", 2e-6)


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_forward_is_bitwise_reproducible(precision):
    """Same weights, same spectral-norm state, same input -> identical bits (the range-tracking
    atomics only form maxima, the convs have a fixed summation order)."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    x = torch.randn(2, 5, 2, 40, 56, generator=torch.Generator().manual_seed(1)).cuda()
    outs = []
    for _ in range(3):
        m = V2ce3d(precision=precision)
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        outs.append(m.eval().to("cuda")(x).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_precision_report_vs_f64_truth():
    """Error of every arithmetic path against an f64 evaluation of the same network (346x260, L=2):
    the CPU f32 restatement (what the reference computes), the exact-f32 HIP path and the split-half
    HIP path.  All must sit inside the 1e-5 bar; the numbers go to gpurun_out/ for profiles/."""
    import json
    sd = synth.make_state_dict(0)
    fr = synth.synthetic_frames(3, 260, 346, seed=1)
    x = fr.astype(np.float32) / 255
    x = np.stack([x[:-1], x[1:]], axis=1)
    x = ((x - np.float32(0.153)) / np.float32(0.165))[None]
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in U.clone_state(sd).items()}
    truth = U.forward(sd64, torch.from_numpy(x).double()).numpy()
    cpu32 = U.forward(U.clone_state(sd), torch.from_numpy(x)).numpy().astype(np.float64)
    rep = {"shape": list(truth.shape), "max_abs_truth": float(np.abs(truth).max())}

    def stats(a):
        d = np.abs(a - truth)
        return {"max_abs": float(d.max()), "rms": float(np.sqrt((d ** 2).mean())),
                "max_excess_over_1e-5_bar": float((d - TOL * np.abs(truth)).max())}

    rep["cpu_f32_oracle"] = stats(cpu32)
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    for prec in ("f32", "f16x2"):
        m = V2ce3d(precision=prec)
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        m = m.eval().to("cuda")
        got = m(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        rep["hip_" + prec] = stats(got)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/precision_report.json", "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    for k in ("cpu_f32_oracle", "hip_f32", "hip_f16x2"):
        assert rep[k]["max_excess_over_1e-5_bar"] <= TOL, (k, rep[k])


def test_batched_spectral_norm_equals_per_layer_calls():
    """v2ce_sn_update_batch (all 12 layers in six launches) against 12 x (v2ce_sn_power_iter +
    v2ce_pack_weights_f16x2): same u / v / packed weights / outputs, bit for bit, over three calls."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    x = torch.from_numpy(OG.preprocess(synth.synthetic_frames(17, 32, 48, seed=6))[None]).cuda()
    ms = []
    for batched in (True, False):
        m = V2ce3d(precision="f16x2")
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        m = m.eval().to("cuda")
        m._prepare()
        assert m._prep["sn_batch"] is not None
        if not batched:
            m._prep["sn_batch"] = None
        ms.append(m)
    for _ in range(3):
        ya, yb = ms[0](x), ms[1](x)
        assert torch.equal(ya, yb)
    for (n, p), (_, q) in zip(ms[0].named_parameters(), ms[1].named_parameters()):
        if n.endswith(("weight_u", "weight_v")):
            assert torch.equal(p, q), n
    for k in ("res0", "res1", "dec0", "dec3"):
        for cn in ("conv1_w", "conv2_w"):
            assert torch.equal(ms[0]._prep[k][cn], ms[1]._prep[k][cn]), (k, cn)


def test_spectral_norm_packed_once_equals_repack(monkeypatch):
    """Round 6: the planes of W_bar packed once, 1 / sigma in the epilogue scale and the folded shortcuts re-packed as Wd' sigma
    (the default) against the re-pack of all twelve layers in every forward (V2CE_SN_REPACK=1, rounds 1-5): the same u / v bit
    for bit, outputs within 2e-6 of each other over three calls (one f32 rounding moves from every weight to every output) and
    both inside the 1e-5 bar against the oracle at every call; the epilogue scales are bn's over the layer's sigma; a
    fast-forwarded replica (advance_spectral_norm) lands on the same state."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    xn = OG.preprocess(synth.synthetic_frames(17, 32, 48, seed=6))[None]
    x = torch.from_numpy(xn).cuda()
    ms = []
    for repack in ("0", "1"):
        monkeypatch.setenv("V2CE_SN_REPACK", repack)
        m = V2ce3d(precision="f16x2")
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        m = m.eval().to("cuda")
        m._prepare()
        ms.append(m)
    monkeypatch.delenv("V2CE_SN_REPACK")
    assert ms[0]._prep["sn_once"] is not None and ms[1]._prep["sn_once"] is None
    assert len(ms[0]._prep["sn_once"]["tails"]) >= 5
    sd = U.clone_state(synth.make_state_dict(0))
    for call in range(3):
        ya, yb = ms[0](x).cpu().numpy(), ms[1](x).cpu().numpy()
        want = U.forward(sd, torch.from_numpy(xn)).numpy()
        assert np.abs(ya - yb).max() <= 2e-6 * max(1.0, np.abs(yb).max()), (call, np.abs(ya - yb).max())
        for y in (ya, yb):
            assert np.all(np.abs(y - want) <= TOL + TOL * np.abs(want)), (call, np.abs(y - want).max())
    for (n, p), (_, q) in zip(ms[0].named_parameters(), ms[1].named_parameters()):
        if n.endswith(("weight_u", "weight_v")):
            assert torch.equal(p, q), n
    d = ms[0]._prep["dec0"]
    sig = 1.0 / ms[0]._prep["sn_once"]["inv_sigma"].cpu().numpy()
    assert np.all(sig > 0) and np.allclose(d["bn2_eff"][0].cpu().numpy() * sig[5], d["bn2"][0].cpu().numpy(), rtol=3e-7)
    # a replica that skips two calls and catches up
    monkeypatch.delenv("V2CE_SN_REPACK", raising=False)
    r = V2ce3d(precision="f16x2")
    r.load_state_dict(synth.make_state_dict(0), strict=True)
    r = r.eval().to("cuda")
    for _ in range(2):
        r.advance_spectral_norm()
    y3 = r(x)
    assert torch.equal(y3, torch.from_numpy(ya).cuda())


def _padded(x_ncdhw, pitch):
    """[N,C,D,H,W] CPU tensor -> device [B,T,C,H,pitch] with NaN in the padding columns and .lw = W."""
    x = to_btchw(x_ncdhw)
    out = torch.full(tuple(x.shape[:-1]) + (pitch,), float("nan"))
    out[..., :x.shape[-1]] = x
    out = out.cuda()
    out.lw = x.shape[-1]
    return out


@pytest.mark.parametrize("split", [False, True])
def test_conv3d_row_pitch(split):
    """Activations with a row pitch > width (v2ce_conv3d_desc.W0_pitch / Win_pitch / Wout_pitch): sources,
    residual and output padded to 96-float rows, NaN in every padding column (never read), virtual
    upsample + concat, 3x3x3 stride 1 and stride 2, 1x1x1; values equal the dense launch bit for bit."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    g = torch.Generator().manual_seed(31)
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2" if split else "f32", 0
    m._prep = {"absmax": torch.zeros((16, 2), device="cuda")}
    for (c0, c1, cout, k, s, use_res) in ((32, 0, 64, 3, 1, True), (32, 16, 32, 3, 1, False), (32, 0, 64, 3, 2, False),
                                          (32, 16, 64, 1, 1, False)):
        H, W = 9, 70
        x0 = torch.randn(1, c0, 3, (H + 1) // 2 if c1 else H, (W + 1) // 2 if c1 else W, generator=g)
        x1 = torch.randn(1, c1, 3, H, W, generator=g) if c1 else None
        w = torch.randn(cout, c0 + c1, k, k, k, generator=g) * 0.05
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        res = torch.randn(1, cout, 3, Ho, Wo, generator=g) if use_res else None
        scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
        wp = V2ce3d._pack(m, w.cuda().contiguous(), split=split)
        outs = []
        for padded in (False, True):
            m._slot = 0
            m._prep["absmax"].zero_()
            if padded:
                a0 = _padded(x0, 96 if not c1 else 64)
                a1 = None if x1 is None else _padded(x1, 96)
                r = None if res is None else _padded(res, V2ce3d._pitch(Wo))
            else:
                a0, a1 = to_btchw(x0).cuda(), None if x1 is None else to_btchw(x1).cuda()
                r = None if res is None else to_btchw(res).cuda()
            if split:
                a0.absmax = torch.nan_to_num(a0).abs().max().reshape(1)
                if a1 is not None:
                    a1.absmax = torch.nan_to_num(a1).abs().max().reshape(1)
                a0, a1, r = (None if v is None else V2ce3d.to_c16(v) for v in (a0, a1, r))   # padding columns travel along
            y = V2ce3d._conv(m, a0, a1, wp, scale, shift, cout, k, s, hip.ACT_RELU, residual=r,
                             up_to=(H, W) if c1 else None, split=split, track=True, dense_out=not padded)
            assert y.shape[4] == (V2ce3d._pitch(Wo) if padded else Wo) and y.lw == Wo
            outs.append((V2ce3d.to_planar(y).contiguous(), float(y.absmax[0])))
        assert not torch.isnan(outs[1][0]).any()
        assert torch.equal(outs[0][0], outs[1][0]), (c0, c1, cout, k, s)
        assert outs[0][1] == outs[1][1] == float(outs[0][0].abs().max())


@pytest.mark.parametrize("case", [(32, 64, 2, 19, 23), (64, 128, 2, 9, 14), (96, 32, 1, 12, 40), (256, 512, 2, 7, 9)])
def test_conv3d_fused_shortcut_vs_f64(case):
    """v2ce_conv3d_fwd_sc: the block's conv1 (3x3x3, BN, ReLU) and its 1x1x1 shortcut (BN, no activation) from
    one launch, against the two convolutions evaluated separately in f64."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    cin, cout, s, H, W = case
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(2, cin, 3, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (cin * 27)) ** 0.5
    wd = torch.randn(cout, cin, 1, 1, 1, generator=g) * (1.0 / cin) ** 0.5
    sc1, sh1 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    sc2, sh2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((4, 2), device="cuda")}
    xd = to_btchw(x).cuda()
    xd.absmax = xd.abs().max().reshape(1)
    y, ysc = V2ce3d._conv(m, V2ce3d.to_c16(xd), None, V2ce3d._pack(m, w.cuda().contiguous(), split=True),
                          sc1.cuda(), sh1.cuda(), cout, 3, s, hip.ACT_RELU, split=True, dense_out=True,
                          sc=(V2ce3d._pack(m, wd.cuda().contiguous(), split=True), sc2.cuda(), sh2.cuda()))
    torch.cuda.synchronize()
    assert_close(V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy(), ref_conv(x, w, sc1, sh1, 3, s, 1), "conv1")
    assert_close(V2ce3d.to_planar(ysc).permute(0, 2, 1, 3, 4).cpu().numpy(), ref_conv(x, wd, sc2, sh2, 1, s, 0), "shortcut")


@pytest.mark.parametrize("case", [
    # T, H, W, with a residual, head channels
    (5, 37, 50, True, 20),        # ragged in T, H and W: partial tiles, masked lanes in both halves of the shared epilogue
    (16, 20, 44, False, 20),      # no residual (the RES = 0 instantiation)
    (3, 9, 70, True, 32),         # every head row present
    (2, 260, 346, True, 20),      # the network's geometry: (8, 4, 16) boxes
])
def test_conv3d_fused_head_vs_f64(case):
    """v2ce_conv3d_fwd_pred: a 32-channel 3x3x3 conv (BN, residual, ReLU) with the 1x1x1 head (bias, ReLU) on the same launch --
    since round 6 with its epilogue shared between consumer and producer waves -- against both layers in f64."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    T, H, W, with_res, pc = case
    B = 2
    g = torch.Generator().manual_seed(T + H + W)
    x = torch.randn(B, 32, T, H, W, generator=g)
    w = torch.randn(32, 32, 3, 3, 3, generator=g) * (2.0 / (32 * 27)) ** 0.5
    sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.3
    res = torch.randn(B, 32, T, H, W, generator=g) if with_res else None
    wp = torch.randn(pc, 32, generator=g) * 0.2
    bp = torch.randn(pc, generator=g) * 0.1
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((4, 2), device="cuda")}

    def c16(t):                                   # channels-last-16 with the activations' row pitch (rows of >= 64 are padded to 32)
        d = V2ce3d.to_c16(to_btchw(t).cuda())
        Wp = V2ce3d._pitch(W)
        if Wp != W:
            p = torch.zeros((*d.shape[:4], Wp, 16), device="cuda")
            p[:, :, :, :, :W] = d
            p.lw, p.c16 = W, True
            d = p
        return d
    xd = c16(x)
    xd.absmax = x.abs().max().reshape(1).cuda()
    rd = c16(res) if with_res else None
    tab = torch.empty(hip.lib().v2ce_pack_pred_weights_f16x2_bytes() // 2, dtype=torch.float16, device="cuda")
    wpd = wp.cuda().contiguous()
    hip.check(hip.lib().v2ce_pack_pred_weights_f16x2(wpd.data_ptr(), pc, 32, tab.data_ptr(), hip.stream_ptr(wpd.device)), "pack")
    bias = torch.zeros(32, device="cuda")
    bias[:pc] = bp.cuda()
    got = V2ce3d._conv(m, xd, None, V2ce3d._pack(m, w.cuda().contiguous(), split=True), sc.cuda(), sh.cuda(), 32, 3, 1, hip.ACT_RELU,
                       residual=rd, split=True, pred=(tab, bias, pc))
    torch.cuda.synchronize()
    assert got.shape == (B, T, pc, H, W)
    y = torch.from_numpy(ref_conv(x, w, sc, sh, 3, 1, 1, residual=res))                      # [B, 32, T, H, W] f64
    want = torch.relu(torch.einsum("oc,bcthw->bothw", wp.double(), y) + bp.double().view(1, -1, 1, 1, 1))
    assert_close(got.permute(0, 2, 1, 3, 4).cpu().numpy(), want.numpy(), "fused head")


@pytest.mark.parametrize("case", [
    # Cmid (= Cout), tail C0, tail C1, tail stride, H, W, mapped low-res source, per-element slots
    (64, 32, 0, 2, 19, 23, False, False),       # encoder-like: strided shortcut from the block input
    (64, 128, 64, 1, 20, 26, True, True),       # decoder-like: shortcut reads upsample(x0) ++ skip (dec2's shape family)
    (128, 256, 128, 1, 9, 13, True, True),      # 128 channels: 256-position boxes
    (256, 256, 0, 1, 17, 22, False, True),      # res-block-like on the 17x22 planes (192-position boxes)
])
def test_conv3d_folded_tail_vs_f64(case):
    """v2ce_conv3d_fwd_tail: relu(s2 (W2 * t + Wd' * x) + shift) from ONE accumulator -- a residual block's conv2
    with the 1x1x1 shortcut folded into its K loop (no shortcut tensor) -- against the two convolutions in f64.  The tail input
    is 1000x the main input's magnitude in one case each way, so the power-of-two rescale between the two parts is exercised."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    cm, c0, c1, ts, H, W, mapped, per_elem = case
    B, T = 2, 3
    g = torch.Generator().manual_seed(cm + c0 + c1)
    t_in = torch.randn(B, cm, T, H, W, generator=g)
    Hx, Wx = (H * ts - (ts - 1), W * ts - (ts - 1)) if ts == 2 else (H, W)
    lo = ((Hx + 1) // 2, (Wx + 1) // 2) if mapped else (Hx, Wx)
    x0 = torch.randn(B, c0, T, *lo, generator=g) * (1000.0 if cm == 64 and ts == 1 else 1.0)
    x1 = torch.randn(B, c1, T, Hx, Wx, generator=g) * 1000.0 if c1 and cm == 64 else (torch.randn(B, c1, T, Hx, Wx, generator=g) if c1 else None)
    if cm == 256:
        t_in = t_in * 300.0
    w2 = torch.randn(cm, cm, 3, 3, 3, generator=g) * (2.0 / (cm * 27)) ** 0.5
    wd = torch.randn(cm, c0 + c1, 1, 1, 1, generator=g) * (1.0 / (c0 + c1)) ** 0.5
    s2, sh = torch.rand(cm, generator=g) + 0.5, torch.randn(cm, generator=g)
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((4, B, 2) if per_elem else (4, 2), device="cuda")}

    def dev(x):
        d = V2ce3d.to_c16(to_btchw(x).cuda())
        d.absmax = (to_btchw(x).abs().amax(dim=(1, 2, 3, 4)).reshape(B, 1).repeat(1, 2).contiguous() if per_elem
                    else x.abs().max().reshape(1)).cuda()
        return d
    y = V2ce3d._conv(m, dev(t_in), None, V2ce3d._pack(m, w2.cuda().contiguous(), split=True), s2.cuda(), sh.cuda(), cm, 3, 1,
                     hip.ACT_RELU, split=True, dense_out=True,
                     tail=(dev(x0), None if x1 is None else dev(x1), (Hx, Wx) if mapped else None, ts,
                           V2ce3d._pack(m, wd.cuda().contiguous(), split=True)))
    torch.cuda.synchronize()
    xs = U.upsample_nearest_hw(x0, (Hx, Wx)).double() if mapped else x0.double()
    if x1 is not None:
        xs = torch.cat([xs, x1.double()], dim=1)
    acc = F.conv3d(t_in.double(), w2.double(), None, 1, 1) + F.conv3d(xs, wd.double(), None, (1, ts, ts), 0)
    want = torch.relu(acc * s2.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)).numpy()
    got = V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()
    # the bar scales with the magnitude of the two parts (the 1000x cases reach |y| ~ 1e3: 1e-5 relative to that)
    scale = max(1.0, float(np.abs(want).max()) / 4.0)
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= TOL * scale + TOL * np.abs(want)), float(np.abs(got - want).max())
    assert 0 < float(y.absmax.reshape(-1, 2)[:, 1].max()) < 1.0        # a finite range-guard bound was reported


def test_folded_shortcut_equals_separate_launches():
    """The whole model with the shortcuts of res0-1 / dec0-2 folded into conv2 (default) vs the same model with the separate
    1x1x1 launches (V2CE_FOLD_SHORTCUT=0): both within the bar of the oracle, and within 2e-6 of each other."""
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    x = torch.from_numpy(np.stack([OG.preprocess(synth.synthetic_frames(17, 40, 56, seed=70 + s)) for s in range(2)])).cuda()
    outs = []
    for fold in ("1", "0"):
        os.environ["V2CE_FOLD_SHORTCUT"] = fold
        try:
            m = V2ce3d()
            m.load_state_dict(synth.make_state_dict(0))
            m = m.eval().to("cuda")
            outs.append(m(x).cpu())
            folded = [k for k, v in m._prep.items() if isinstance(v, dict) and v.get("fold") is not None]
            assert folded == (["res0", "res1", "dec0", "dec1", "dec2"] if fold == "1" else [])
        finally:
            os.environ.pop("V2CE_FOLD_SHORTCUT", None)
    want = U.forward(U.clone_state(synth.make_state_dict(0)), x.cpu()).contiguous()
    for o in outs:
        assert_close(o.numpy(), want.numpy(), "vs oracle")
    assert float((outs[0] - outs[1]).abs().max()) < 2e-6
