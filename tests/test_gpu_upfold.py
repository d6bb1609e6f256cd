"""GPU parity of the phase-folded decoder convolution (csrc/conv3d_up.hip, v2ce_conv3d_fwd_up2): conv1 of a decoder
block -- nearest-upsample-2x(x0) ++ skip -> 3x3x3 conv + BN + ReLU (+ the fused 1x1x1 shortcut), reference
/root/reference/scripts/unet_2layer.py:358-365 and /root/reference/scripts/submodules.py:249-264 -- against the same
convolution evaluated in f64 on the materialised upsample + concat, and against the generic (mapped gather, 27 taps)
kernel on the same buffers.  Tolerance (north_star): 1e-5 abs + 1e-5 rel."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet as U

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


def ref_conv(x0, w, scale, shift, ksize, stride, act, x1=None, up_to=None):
    """f64 evaluation of conv(upsample_nearest(x0) ++ x1) * scale + shift, activation; NCDHW."""
    x = x0.double()
    if up_to is not None:
        x = U.upsample_nearest_hw(x0, up_to).double()
    if x1 is not None:
        x = torch.cat([x, x1.double()], dim=1)
    y = F.conv3d(x, w.double(), None, (1, stride, stride), ksize // 2)
    y = y * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if act == 1:
        y = torch.relu(y)
    return y.numpy()


def _model():
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((8, 2), device="cuda")}
    return m


def _up_weights(m, w, c0, sigma=None):
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    cout, cin = w.shape[0], w.shape[1]
    buf = V2ce3d._split_buffer(cout, cin, 27, "cuda", up_c0=c0)
    return V2ce3d._pack(m, w.cuda().contiguous(), sigma, buf, split=True)


UP_CASES = [
    # B, T, C0, C1, Cout, Hout, Wout, fused shortcut          (x0 is ceil(Hout / 2) x ceil(Wout / 2))
    (1, 3, 64, 32, 32, 20, 28, True),        # dec3 family: even x even, shortcut, one phase per wave
    (2, 5, 64, 32, 32, 21, 27, True),        # odd x odd with the shortcut: every correction list, ragged T
    (1, 4, 128, 64, 64, 26, 35, False),      # dec2 family: W odd (64 channels x 512 positions)
    (1, 16, 256, 128, 128, 17, 23, False),   # dec1 family: both odd, two phases per wave
    (2, 16, 512, 256, 256, 9, 12, False),    # dec0 family: H odd, two channel tiles
    (1, 2, 32, 16, 32, 2, 2, False),         # a single source pixel
    (1, 1, 16, 16, 64, 1, 5, False),         # one row (H0 = 1): only the "last row" exists
    (1, 3, 32, 32, 128, 33, 7, False),       # tall and narrow, both odd
]


@pytest.mark.parametrize("case", UP_CASES)
def test_conv3d_up2_vs_f64(case):
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, C0, C1, Cout, H, W, with_sc = case
    H0, W0 = (H + 1) // 2, (W + 1) // 2
    g = torch.Generator().manual_seed(C0 + 3 * H + W)
    # (K = 27 (C0 + C1) products per output, accumulated in f32 by the MFMA like the reference's own f32 convolution: at
    # K = 20 736 and outputs of O(1) that noise alone reaches 1e-5, so the large-K cases run on inputs of O(0.25))
    amp = 0.25 if (C0 + C1) * 27 > 8000 else 1.0
    x0 = amp * torch.randn(B, C0, T, H0, W0, generator=g)
    x1 = amp * torch.randn(B, C1, T, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) * (2.0 / ((C0 + C1) * 27)) ** 0.5
    sc1, sh1 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    m = _model()
    xd0, xd1 = to_btchw(x0).cuda(), to_btchw(x1).cuda()
    xd0.absmax, xd1.absmax = xd0.abs().max().reshape(1), xd1.abs().max().reshape(1)
    wq = _up_weights(m, w, C0)
    sc = None
    if with_sc:
        wd = torch.randn(Cout, C0 + C1, 1, 1, 1, generator=g) * (1.0 / (C0 + C1)) ** 0.5
        sc2, sh2 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        sc = (V2ce3d._pack(m, wd.cuda().contiguous(), split=True), sc2.cuda(), sh2.cuda())
    m.profile = []
    out = V2ce3d._conv(m, V2ce3d.to_c16(xd0), V2ce3d.to_c16(xd1), wq, sc1.cuda(), sh1.cuda(), Cout, 3, 1, hip.ACT_RELU,
                       up_to=(H, W), split=True, dense_out=True, sc=sc)
    torch.cuda.synchronize()
    assert "conv3d_up_kernel" in m.profile[0][0], m.profile[0][0]          # the folded kernel ran, not the mapped gather
    y = out[0] if with_sc else out
    assert_close(V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy(),
                 ref_conv(x0, w, sc1, sh1, 3, 1, 1, x1=x1, up_to=(H, W)), f"conv1 {case}")
    if with_sc:
        assert_close(V2ce3d.to_planar(out[1]).permute(0, 2, 1, 3, 4).cpu().numpy(),
                     ref_conv(x0, wd, sc2, sh2, 1, 1, 0, x1=x1, up_to=(H, W)), f"shortcut {case}")


def test_up_buffer_is_a_plain_buffer_too():
    """The plain planes of a v2ce_pack_weights_f16x2_up buffer drive the generic kernel (mapped gather, 27 taps); the two
    kernels agree inside the parity bar (they differ by the order of their f32 accumulations -- 2592 products per output
    here -- and by the pre-summed weights)."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, C0, C1, Cout, H, W = 1, 4, 64, 32, 64, 13, 18
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, C0, T, 7, 9, generator=g)
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) * 0.03
    sc1, sh1 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    m = _model()
    xd0, xd1 = to_btchw(x0).cuda(), to_btchw(x1).cuda()
    xd0.absmax, xd1.absmax = xd0.abs().max().reshape(1), xd1.abs().max().reshape(1)
    wq = _up_weights(m, w, C0)
    args = (V2ce3d.to_c16(xd0), V2ce3d.to_c16(xd1), None, sc1.cuda(), sh1.cuda(), Cout, 3, 1, hip.ACT_RELU)
    m.profile = []
    y_up = V2ce3d._conv(m, *args[:2], wq, *args[3:], up_to=(H, W), split=True, dense_out=True)
    plain = wq[:]                           # same storage, without the attribute that selects the folded kernel
    y_gen = V2ce3d._conv(m, *args[:2], plain, *args[3:], up_to=(H, W), split=True, dense_out=True)
    torch.cuda.synchronize()
    assert "conv3d_up_kernel" in m.profile[0][0] and "ws_kernel" in m.profile[1][0], [p[0] for p in m.profile]
    a, b = V2ce3d.to_planar(y_up).cpu().numpy(), V2ce3d.to_planar(y_gen).cpu().numpy()
    assert_close(a, b, "folded vs mapped")
    want = ref_conv(x0, w, sc1, sh1, 3, 1, 1, x1=x1, up_to=(H, W))
    assert_close(np.transpose(b, (0, 2, 1, 3, 4)), want, "mapped kernel on the up buffer")


def test_up_pack_with_sigma_and_range():
    """v2ce_pack_weights_f16x2_up: the common pre-scale covers the folded sums (up to four weights of one sign: no fp16
    overflow), sigma divides before the sums, and a weight tensor spanning many binades still gives f32-grade results."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, C0, C1, Cout, H, W = 1, 2, 32, 16, 32, 9, 11
    g = torch.Generator().manual_seed(23)
    x0 = torch.randn(B, C0, T, 5, 6, generator=g)
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    w = torch.full((Cout, C0 + C1, 3, 3, 3), 0.75)               # every folded sum is 2x / 4x a single weight
    w += 1e-3 * torch.randn(w.shape, generator=g)
    w[:, ::3] *= 1e-4                                             # some channels 13 binades down
    sigma = torch.tensor([3.7])
    sc1, sh1 = torch.ones(Cout), torch.zeros(Cout)
    m = _model()
    xd0, xd1 = to_btchw(x0).cuda(), to_btchw(x1).cuda()
    xd0.absmax, xd1.absmax = xd0.abs().max().reshape(1), xd1.abs().max().reshape(1)
    wq = _up_weights(m, w, C0, sigma.cuda())
    tail = wq[Cout * (C0 + C1) * 27 * 2:][:8].view(torch.float32).cpu().numpy()
    folded_max = float((4 * w[:, :C0].abs().max() / sigma).item())
    assert tail[0] >= 0.98 * folded_max and tail[0] * tail[1] < 32768 and tail[2] == 0 and tail[3] == 0, tail
    y = V2ce3d._conv(m, V2ce3d.to_c16(xd0), V2ce3d.to_c16(xd1), wq, sc1.cuda(), sh1.cuda(), Cout, 3, 1, hip.ACT_NONE,
                     up_to=(H, W), split=True, dense_out=True)
    torch.cuda.synchronize()
    want = ref_conv(x0, w / sigma, sc1, sh1, 3, 1, 0, x1=x1, up_to=(H, W))
    assert_close(V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy(), want, "sigma + wide range")


def test_up2_rejects_what_it_cannot_do():
    from v2ce_toolbox_amd import hip
    L = hip.lib()
    d = hip.ConvDesc(B=1, T=2, C0=32, H0=5, W0=6, C1=16, Hin=9, Win=11, Cout=32, Hout=9, Wout=11, ksize=3, stride_hw=1,
                     act=1, tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2, W0_pitch=0, Win_pitch=0,
                     Wout_pitch=0, layout=hip.LAYOUT_C16, absmax_batch_stride=0)
    buf = ctypes.create_string_buffer(96)
    assert L.v2ce_conv3d_up2_variant(ctypes.byref(d), 0, buf, 96) == 0 and b"conv3d_up_kernel" in buf.value
    d.H0 = 4                                   # not the 2x source of a 9-row output
    assert L.v2ce_conv3d_up2_variant(ctypes.byref(d), 0, buf, 96) == -1
    d.H0, d.layout = 5, hip.LAYOUT_PLANAR
    assert L.v2ce_conv3d_up2_variant(ctypes.byref(d), 0, buf, 96) == -2
    d.layout, d.C1 = hip.LAYOUT_C16, 0
    assert L.v2ce_conv3d_up2_variant(ctypes.byref(d), 0, buf, 96) == -1


# ------------------------------------------------------------------------------------------------
# the head convolution in split-half arithmetic (csrc/conv3d_head.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(1, 4, 20, 28), (2, 5, 37, 71), (1, 3, 9, 33), (2, 16, 8, 130), (1, 1, 1, 1)])
def test_head_split_half_vs_f64(case):
    """v2ce_conv3d_head_f16x2 (Conv3d(2, 32, 3, padding 1) + bias + LeakyReLU, unet_2layer.py:341 / submodules.py:115-124)
    against the f64 convolution: ragged in T, H and W (boxes of 4 x 4 x 64), inputs spanning the range of normalised frames
    (-0.93 .. 5.1), per-sequence range slots; max |y| and the range-guard value are reported."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, H, W = case
    g = torch.Generator().manual_seed(7 + H + W)
    x = torch.rand(B, 2, T, H, W, generator=g) * 6.0 - 0.93
    x[0] *= 0.01                                                  # sequence 0 lives 100x lower: its own pre-scale
    w = torch.randn(32, 2, 3, 3, 3, generator=g) * (2.0 / 54) ** 0.5
    bias = 0.3 * torch.randn(32, generator=g)
    m = _model()
    m._prep = {"absmax": torch.zeros((8, B, 2), device="cuda")}
    tab = torch.empty(hip.lib().v2ce_pack_head_weights_f16x2_bytes() // 2, dtype=torch.float16, device="cuda")
    hip.check(hip.lib().v2ce_pack_head_weights_f16x2(w.cuda().contiguous().data_ptr(), tab.data_ptr(), hip.stream_ptr("cuda")), "pack")
    y = V2ce3d._head_split(m, to_btchw(x).cuda(), tab, bias.cuda())
    torch.cuda.synchronize()
    want = F.leaky_relu(F.conv3d(x.double(), w.double(), bias.double(), 1, 1), 0.01).numpy()
    got = V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()
    assert_close(got, want, f"head {case}")
    slots = m._prep["absmax"].cpu().numpy()
    for b in range(B):
        assert abs(slots[0, b, 0] - float(x[b].abs().max())) <= 1e-6 * float(x[b].abs().max())       # max |x| of the sequence
        assert abs(slots[1, b, 0] - np.abs(want[b]).max()) <= 2e-5 * max(1.0, np.abs(want[b]).max())  # max |y|
        assert 0 < slots[1, b, 1] < 2.5e-6                                                           # its guard value


@pytest.mark.parametrize("case", [(1, 4, 128, 64, 64, 26, 35), (2, 16, 256, 128, 128, 17, 23), (1, 5, 64, 32, 64, 21, 27)])
def test_conv3d_up2_split_launch_vs_f64(case):
    """conv1 of a wide decoder block as two launches (v2ce_conv3d_fwd_up2_part: the upsampled channels, phase-folded, no activation;
    v2ce_conv3d_fwd_wt: the skip channels on the Winograd-T kernel with the first launch's output as its residual): the f64
    convolution of upsample(x0) ++ x1, and the partial sum alone."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, C0, C1, Cout, H, W = case
    H0, W0 = (H + 1) // 2, (W + 1) // 2
    g = torch.Generator().manual_seed(C0 + 5 * H + W)
    amp = 0.25 if (C0 + C1) * 27 > 8000 else 1.0
    x0 = amp * torch.randn(B, C0, T, H0, W0, generator=g)
    x1 = amp * torch.randn(B, C1, T, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, 3, generator=g) * (2.0 / ((C0 + C1) * 27)) ** 0.5
    sc1, sh1 = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    sigma = torch.tensor([1.3], device="cuda")
    m = _model()
    xd0, xd1 = to_btchw(x0).cuda(), to_btchw(x1).cuda()
    xd0.absmax, xd1.absmax = xd0.abs().max().reshape(1), xd1.abs().max().reshape(1)
    c0, c1 = V2ce3d.to_c16(xd0), V2ce3d.to_c16(xd1)
    wq = _up_weights(m, w, C0, sigma)
    sk = V2ce3d._split_buffer(Cout, C1, 27, "cuda", wt=True)
    sk.ci0 = C0
    V2ce3d._pack(m, w.cuda().contiguous(), sigma, sk, split=True)
    assert V2ce3d._up2_ok(m, c0, c1, wq, (H, W))
    m.profile = []
    part = V2ce3d._conv_up_part(m, c0, c1, wq, sc1.cuda(), sh1.cuda(), Cout, (H, W))
    y = V2ce3d._conv(m, c1, None, sk, sc1.cuda(), torch.zeros(Cout, device="cuda"), Cout, 3, 1, hip.ACT_RELU, residual=part, split=True)
    torch.cuda.synchronize()
    assert "conv3d_up_kernel" in m.profile[0][0] and "conv3d_wt_kernel" in m.profile[1][0], [p[0] for p in m.profile]
    ws = w / 1.3
    w_up_only = ws.clone()
    w_up_only[:, C0:] = 0
    assert_close(V2ce3d.to_planar(part).permute(0, 2, 1, 3, 4).cpu().numpy(),
                 ref_conv(x0, w_up_only, sc1, sh1, 3, 1, 0, x1=x1, up_to=(H, W)), f"partial sum {case}")
    assert_close(V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy(),
                 ref_conv(x0, ws, sc1, sh1, 3, 1, 1, x1=x1, up_to=(H, W)), f"split conv1 {case}")


@pytest.mark.parametrize("hw", [(32, 48), (33, 47), (65, 90)])
def test_last_decoder_shortcut_split_by_source(hw, monkeypatch):
    """Round 6: dec3 with its 1x1x1 shortcut split by source (plain phase-folded conv1; the upsampled channels' share at the source's
    resolution as conv2's low-resolution residual; the skip channels as conv2's folded tail; pred fused: v2ce_conv3d_fwd_tail_pred)
    -- opt-in, V2CE_DEC3_SPLIT=1: measured slower than the fused form, DESIGN 4.1h -- against the fused-shortcut form of rounds 2-5
    (the default) and against the oracle -- even and odd planes (the residual
    is read at (h >> 1, w >> 1) with ceil(H / 2) rows), three calls (the tail is re-packed as Wd' sigma in every forward)."""
    from oracle import glue as OG
    from oracle import unet as U
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    H, W = hw
    xn = OG.preprocess(synth.synthetic_frames(9, H, W, seed=11))[None]
    x = torch.from_numpy(xn).cuda()
    ms = []
    for split in ("1", "0"):
        monkeypatch.setenv("V2CE_DEC3_SPLIT", split)
        m = V2ce3d(precision="f16x2")
        m.load_state_dict(synth.make_state_dict(0), strict=True)
        m = m.eval().to("cuda")
        m._prepare()
        ms.append(m)
    monkeypatch.delenv("V2CE_DEC3_SPLIT")
    assert ms[0]._prep["dec3"].get("fold_lo") is not None and ms[1]._prep["dec3"].get("fold_lo") is None
    sd = U.clone_state(synth.make_state_dict(0))
    for call in range(3):
        ya, yb = ms[0](x).cpu().numpy(), ms[1](x).cpu().numpy()
        want = U.forward(sd, torch.from_numpy(xn)).numpy()
        assert np.abs(ya - yb).max() <= 3e-6 * max(1.0, np.abs(yb).max()), (call, np.abs(ya - yb).max())
        for y in (ya, yb):
            assert np.all(np.abs(y - want) <= 1e-5 + 1e-5 * np.abs(want)), (call, np.abs(y - want).max())
    ms[0].profile = []
    ms[0](x)
    names = [p[0] for p in ms[0].profile]
    assert any(n.startswith("conv3d_f16x2_ws_kernel<3,1,1,1,4,9,4,1,") for n in names), names
    assert any(n.startswith("conv3d_up_kernel<1,1,4,0>") for n in names), names
