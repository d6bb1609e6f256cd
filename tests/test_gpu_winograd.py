"""GPU parity of the Winograd-T convolution (csrc/conv3d_wt.hip, v2ce_conv3d_fwd_wt): a 3x3x3 stride-1 conv + BN (+ residual)
+ ReLU of a residual block -- reference /root/reference/scripts/submodules.py:249-264 (conv2 of every block, conv1 of the two
middle blocks) -- against the same convolution evaluated in f64, and against the direct split-half kernel on the same buffers.
Tolerance (north_star): 1e-5 abs + 1e-5 rel."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

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


def ref_conv(x, w, scale, shift, act, residual=None):
    y = F.conv3d(x.double(), w.double(), None, 1, 1)
    y = y * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if residual is not None:
        y = y + residual.double()
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


def _weights(m, w, wt, sigma=None):
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    buf = V2ce3d._split_buffer(w.shape[0], w.shape[1], 27, "cuda", wt=wt)
    return V2ce3d._pack(m, w.cuda().contiguous(), sigma, buf, split=True)


CASES = [
    # B, T, Cin, Cout, H, W, residual
    (1, 4, 64, 64, 12, 20, False),
    (2, 16, 64, 64, 33, 44, True),        # enc0 / dec2 family, whole pairs
    (1, 5, 32, 64, 9, 7, True),           # odd T: the last pair's second step does not exist
    (1, 1, 16, 64, 5, 5, False),          # a single time step
    (1, 16, 128, 128, 17, 22, True),      # two channel tiles
    (2, 16, 256, 256, 9, 11, True),       # res family: four channel tiles, K = 6912
    (1, 6, 64, 64, 1, 40, False),         # one row
    (1, 2, 48, 192, 40, 3, True),         # three columns, three channel tiles
]


@pytest.mark.parametrize("case", CASES)
def test_conv3d_wt_vs_f64(case):
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, Cin, Cout, H, W, with_res = case
    g = torch.Generator().manual_seed(Cin + 3 * H + W)
    amp = 0.25 if Cin * 27 > 6000 else 1.0
    x = amp * torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = amp * torch.randn(B, Cout, T, H, W, generator=g) if with_res else None
    m = _model()
    xd = to_btchw(x).cuda()
    xd.absmax = xd.abs().max().reshape(1)
    wq = _weights(m, w, True)
    rd = V2ce3d.to_c16(to_btchw(res).cuda()) if with_res else None
    m.profile = []
    y = V2ce3d._conv(m, V2ce3d.to_c16(xd), None, wq, sc.cuda(), sh.cuda(), Cout, 3, 1, hip.ACT_RELU, residual=rd, split=True,
                     dense_out=True, track=True)
    torch.cuda.synchronize()
    assert "conv3d_wt_kernel" in m.profile[0][0], m.profile[0][0]
    want = ref_conv(x, w, sc, sh, 1, res)
    got = V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()
    assert_close(got, want, f"wt {case}")
    # range tracking: max |y| of the launch, and a finite guard bound
    am = y.absmax.cpu().numpy().reshape(-1)
    assert abs(am[0] - np.abs(got).max()) <= 1e-6 * max(1.0, am[0]) and np.isfinite(am[1]) and am[1] > 0


def test_wt_equals_direct_kernel_inside_the_bar():
    """The same tensors through the direct split-half kernel (27 taps) and the Winograd-T one (36 transformed taps per pair):
    different summation orders and one more f32 rounding per transformed operand, inside the parity bar; spectral-norm
    sigma is divided out before the transform."""
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, Cin, Cout, H, W = 2, 8, 64, 128, 21, 30
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.05
    sigma = torch.tensor([1.7], device="cuda")
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    m = _model()
    xd = to_btchw(x).cuda()
    xd.absmax = xd.abs().max().reshape(1)
    xc = V2ce3d.to_c16(xd)
    ya = V2ce3d._conv(m, xc, None, _weights(m, w, True, sigma), sc.cuda(), sh.cuda(), Cout, 3, 1, hip.ACT_RELU, split=True, dense_out=True)
    yb = V2ce3d._conv(m, xc, None, _weights(m, w, False, sigma), sc.cuda(), sh.cuda(), Cout, 3, 1, hip.ACT_RELU, split=True, dense_out=True)
    torch.cuda.synchronize()
    assert_close(ya.cpu().numpy(), yb.cpu().numpy(), "winograd-T vs direct")
    assert_close(V2ce3d.to_planar(ya).permute(0, 2, 1, 3, 4).cpu().numpy(), ref_conv(x, w / 1.7, sc, sh, 1), "winograd-T vs f64")


def test_wt_rejects_what_it_cannot_do():
    from v2ce_toolbox_amd import hip
    import ctypes
    d = hip.ConvDesc(B=1, T=2, C0=16, H0=4, W0=4, C1=0, Hin=4, Win=4, Cout=32, Hout=4, Wout=4, ksize=3, stride_hw=1, act=1,
                     tile_t=0, tile_h=0, tile_w=0, precision=hip.PRECISION_F16X2, W0_pitch=4, Win_pitch=4, Wout_pitch=4,
                     layout=hip.LAYOUT_C16, absmax_batch_stride=0)
    t = torch.zeros(4096, device="cuda")
    rc = hip.lib().v2ce_conv3d_fwd_wt(ctypes.byref(d), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, t.data_ptr(),
                                      None, None, None)
    assert rc != 0 and b"Cout" in hip.lib().v2ce_last_error()


@pytest.mark.parametrize("case", [
    # Cmid (= Cout), tail C0, tail C1, tail stride, H, W, mapped low-res source, per-element slots, T
    (64, 64, 0, 2, 19, 23, False, False, 3),      # strided shortcut from the block input
    (64, 128, 64, 1, 20, 26, True, True, 3),      # decoder-like: shortcut reads upsample(x0) ++ skip (dec2's shape family), odd T
    (128, 256, 128, 1, 9, 13, True, True, 4),     # dec1's family
    (256, 512, 256, 1, 8, 11, True, False, 2),    # dec0's family: 768 tail channels over two sources
    (256, 256, 0, 1, 17, 22, False, True, 16),    # res-block-like on the 17x22 planes (flat range tiles)
])
def test_conv3d_wt_folded_tail_vs_f64(case):
    """v2ce_conv3d_fwd_wt_tail: relu(s2 (W2 * t + Wd' * x) + shift) from ONE accumulator set in the transform domain -- a residual
    block's conv2 on the Winograd-T kernel with the 1x1x1 shortcut folded into its K loop -- against the two convolutions in f64
    (same cases and bar as tests/test_gpu_unet.py::test_conv3d_folded_tail_vs_f64, incl. the 1000x magnitude gap both ways)."""
    from oracle import unet as U
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    cm, c0, c1, ts, H, W, mapped, per_elem, T = case
    B = 2
    g = torch.Generator().manual_seed(cm + c0 + c1)
    t_in = torch.randn(B, cm, T, H, W, generator=g)
    Hx, Wx = (H * ts - (ts - 1), W * ts - (ts - 1)) if ts == 2 else (H, W)
    lo = ((Hx + 1) // 2, (Wx + 1) // 2) if mapped else (Hx, Wx)
    x0 = torch.randn(B, c0, T, *lo, generator=g) * (1000.0 if cm == 64 and ts == 1 else 1.0)
    x1 = torch.randn(B, c1, T, Hx, Wx, generator=g) * 1000.0 if c1 and cm == 64 else (torch.randn(B, c1, T, Hx, Wx, generator=g) if c1 else None)
    if cm == 256 and c1 == 0:
        t_in = t_in * 300.0
    w2 = torch.randn(cm, cm, 3, 3, 3, generator=g) * (2.0 / (cm * 27)) ** 0.5
    wd = torch.randn(cm, c0 + c1, 1, 1, 1, generator=g) * (1.0 / (c0 + c1)) ** 0.5
    s2, sh = torch.rand(cm, generator=g) + 0.5, torch.randn(cm, generator=g)
    m = _model()
    m._prep = {"absmax": torch.zeros((4, B, 2) if per_elem else (4, 2), device="cuda")}

    def dev(x):
        d = V2ce3d.to_c16(to_btchw(x).cuda())
        d.absmax = (to_btchw(x).abs().amax(dim=(1, 2, 3, 4)).reshape(B, 1).repeat(1, 2).contiguous() if per_elem
                    else x.abs().max().reshape(1)).cuda()
        return d
    m.profile = []
    y = V2ce3d._conv(m, dev(t_in), None, _weights(m, w2, True), s2.cuda(), sh.cuda(), cm, 3, 1, hip.ACT_RELU, split=True, dense_out=True,
                     tail=(dev(x0), None if x1 is None else dev(x1), (Hx, Wx) if mapped else None, ts,
                           V2ce3d._pack(m, wd.cuda().contiguous(), split=True)))
    torch.cuda.synchronize()
    assert m.profile[0][0] == "conv3d_wt_kernel<2,4,0,1>", m.profile[0][0]
    xs = U.upsample_nearest_hw(x0, (Hx, Wx)).double() if mapped else x0.double()
    if x1 is not None:
        xs = torch.cat([xs, x1.double()], dim=1)
    acc = F.conv3d(t_in.double(), w2.double(), None, 1, 1) + F.conv3d(xs, wd.double(), None, (1, ts, ts), 0)
    want = torch.relu(acc * s2.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)).numpy()
    got = V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()
    scale = max(1.0, float(np.abs(want).max()) / 4.0)
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= TOL * scale + TOL * np.abs(want)), float(np.abs(got - want).max())
    assert 0 < float(y.absmax.reshape(-1, 2)[:, 1].max()) < 1.0        # a finite range-guard bound was reported


def test_conv3d_wt_tail_with_upsampled_residual_vs_f64():
    """A decoder block's conv2 with the folded shortcut split by source: the skip channels ride as the tail of the Winograd-T launch,
    the upsampled channels' share arrives as a LOW-resolution residual read at (h >> 1, w >> 1) (v2ce_conv3d_fwd_wt_tail, res_h) --
    against conv2 + the whole 1x1x1 shortcut in f64.  Odd output size: the last source row / column feeds one output row / column."""
    from oracle import unet as U
    from v2ce_toolbox_amd import hip
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    B, T, cm, c0, c1, H, W = 2, 5, 128, 256, 128, 13, 19
    g = torch.Generator().manual_seed(17)
    t_in = torch.randn(B, cm, T, H, W, generator=g)
    x0 = torch.randn(B, c0, T, (H + 1) // 2, (W + 1) // 2, generator=g)
    x1 = torch.randn(B, c1, T, H, W, generator=g)
    w2 = torch.randn(cm, cm, 3, 3, 3, generator=g) * (2.0 / (cm * 27)) ** 0.5
    wd = torch.randn(cm, c0 + c1, 1, 1, 1, generator=g) * (1.0 / (c0 + c1)) ** 0.5
    s2, sh = torch.rand(cm, generator=g) + 0.5, torch.randn(cm, generator=g)
    m = _model()

    def dev(x):
        d = V2ce3d.to_c16(to_btchw(x).cuda())
        d.absmax = x.abs().max().reshape(1).cuda()
        return d
    lo = V2ce3d._pack(m, wd[:, :c0].contiguous().cuda(), split=True)
    sk = V2ce3d._pack(m, wd[:, c0:].contiguous().cuda(), split=True)
    r0 = V2ce3d._conv(m, dev(x0), None, lo, s2.cuda(), torch.zeros(cm, device="cuda"), cm, 1, 1, hip.ACT_NONE, split=True)
    m.profile = []
    y = V2ce3d._conv(m, dev(t_in), None, _weights(m, w2, True), s2.cuda(), sh.cuda(), cm, 3, 1, hip.ACT_RELU, split=True, dense_out=True,
                     tail=(dev(x1), None, None, 1, sk), residual=r0, residual_up=True)
    torch.cuda.synchronize()
    assert m.profile[0][0] == "conv3d_wt_kernel<2,4,1,1>", m.profile[0][0]
    xs = torch.cat([U.upsample_nearest_hw(x0, (H, W)).double(), x1.double()], dim=1)
    acc = F.conv3d(t_in.double(), w2.double(), None, 1, 1) + F.conv3d(xs, wd.double(), None, 1, 0)
    want = torch.relu(acc * s2.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)).numpy()
    assert_close(V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy(), want, "conv2 + split shortcut")


@pytest.mark.parametrize("case", [(6, 64, 96), (2, 40, 56), (4, 33, 47)])
def test_network_with_odd_sequence_lengths_vs_oracle(case):
    """The whole UNet on sequences of 5 / 1 / 3 frame-pairs (the Winograd-T kernels pair time steps: the last pair's second step does
    not exist; ragged planes) against the oracle at 1e-5 -- reference forward /root/reference/scripts/unet_2layer.py:335-379."""
    from oracle import glue as OG
    from oracle import unet as U
    from v2ce_toolbox_amd import synth
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    nfr, H, W = case
    sd = synth.make_state_dict(0)
    x = OG.preprocess(synth.synthetic_frames(nfr, H, W, seed=5))[None]
    want = U.forward({k: v.clone() for k, v in sd.items()}, torch.from_numpy(x)).contiguous().numpy()
    m = V2ce3d()
    m.load_state_dict(sd)
    m = m.eval().to("cuda")
    m.profile = []
    got = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert any("conv3d_wt_kernel" in p[0] for p in m.profile), "the Winograd-T launches ran"
    assert_close(got, want, f"network {case}")
