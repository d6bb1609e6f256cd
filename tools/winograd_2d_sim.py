"""Paper study (CPU, oracle-side only): would a 2-D Winograd F(2x2, 3x3) on the (T, W) plane -- nested F(2,3) along T
and along W, the H taps direct -- keep the 1e-5 parity bar on the layers it would pay for (>= 256 channels: enc2.conv2, enc3.conv2,
res0-1 conv1 / conv2, dec0.conv2: 2.25x fewer products than direct against 1.5x for the 1-D form the product runs today)?

The product's arithmetic is modelled, not idealised: every operand of a 3x3x3 convolution -- transformed or not -- is rounded to
what the split-half kernels hold (hi + lo fp16 at a power-of-two pre-scale of the tensor's max: 22 bits), the lo x lo product is
dropped, sums and the output transform are f32.  Three arms per weight state, all against the plain f32 oracle (the parity bar)
and an f64 run:
    direct   every conv direct (22-bit operands)
    wino-T   1-D F(2,3) along T on the layers the product runs it on (stride 1, Cin = Cout, Cout % 64 == 0): today's build
    wino-TW  the same, the >= 256-channel ones of them nested along W as well
Not product code.   python tools/winograd_2d_sim.py [H W L]   (default 64 96 16: the deepest maps are 4 x 6)"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import glue as OG  # noqa: E402
from oracle import unet as U  # noqa: E402
from v2ce_toolbox_amd import synth  # noqa: E402

direct = F.conv3d
MODE = {"arm": "oracle"}


def split22(v):
    """v as the split-half kernels hold it: (hi + lo) / s with hi = fp16(v s), lo = fp16(v s - hi), s = the power of two that maps
    max |v| below 4096 (conv3d_dev.h pow2_prescale); returns the 22-bit value and its lo part (for the dropped lo x lo term)."""
    if v.dtype != torch.float32:
        return v, torch.zeros_like(v)
    m = float(v.abs().max())
    s = 2.0 ** np.floor(np.log2(4094.0 / m)) if m > 0 else 1.0
    vs = v * s
    hi = vs.half().float()
    lo = (vs - hi).half().float()
    return (hi + lo) / s, lo / s


def mm(conv, x, w):
    """A product of 22-bit operands without its lo x lo term (three MFMAs per k-step, f32 accumulate)."""
    xq, xl = split22(x)
    wq, wl = split22(w)
    return conv(xq, wq) - conv(xl, wl)


G1 = lambda g0, g1, g2: [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]          # F(2,3) weight transform
B1 = lambda d0, d1, d2, d3: [d0 - d2, d1 + d2, d2 - d1, d1 - d3]                        # input transform
A1 = lambda m: (m[0] + m[1] + m[2], m[1] - m[2] - m[3])                                 # output transform


def conv_any(x, w, b, stride, pad):
    arm = MODE["arm"]
    s1 = (stride == 1) or (not isinstance(stride, int) and tuple(stride) == (1, 1, 1))
    if arm == "oracle" or w.shape[2:] != (3, 3, 3):
        return direct(x, w, b, stride, pad)
    wino = arm != "direct" and s1 and w.shape[0] == w.shape[1] and w.shape[0] % 64 == 0
    if not wino:
        out = mm(lambda a, c: direct(a, c, None, stride, pad), x, w)
        return out if b is None else out + b.view(1, -1, 1, 1, 1)
    two_d = arm == "wino-TW" and w.shape[0] >= 256
    Bn, C, T, H, W = x.shape
    Tp = T + (T & 1)
    if not two_d:
        xp = F.pad(x, (0, 0, 0, 0, 1, 1 + Tp - T))
        Gs = G1(w[:, :, 0], w[:, :, 1], w[:, :, 2])                                     # [Cout, Cin, 3, 3] each
        out = x.new_zeros(Bn, w.shape[0], Tp, H, W)
        for p in range(Tp // 2):
            D = B1(*(xp[:, :, 2 * p + i] for i in range(4)))
            m = [mm(lambda a, c: F.conv2d(a, c, None, 1, 1), D[i], Gs[i]) for i in range(4)]
            out[:, :, 2 * p], out[:, :, 2 * p + 1] = A1(m)
        out = out[:, :, :T]
    else:
        Wp = W + (W & 1)
        xp = F.pad(x, (1, 1 + Wp - W, 0, 0, 1, 1 + Tp - T))                             # T and W padded, H by the conv below
        Gt = G1(w[:, :, 0], w[:, :, 1], w[:, :, 2])                                     # along T: [Cout, Cin, 3(h), 3(w)]
        U2 = [[None] * 4 for _ in range(4)]
        for i in range(4):
            gw = G1(Gt[i][..., 0], Gt[i][..., 1], Gt[i][..., 2])                        # along W: [Cout, Cin, 3(h)]
            for j in range(4):
                U2[i][j] = gw[j]
        out = x.new_zeros(Bn, w.shape[0], Tp, H, Wp)
        nW = Wp // 2
        for p in range(Tp // 2):
            Dt = B1(*(xp[:, :, 2 * p + i] for i in range(4)))                           # [B, C, H, Wp + 2] each
            M = [[None] * 4 for _ in range(4)]
            for i in range(4):
                cols = [Dt[i][..., k::2][..., :nW + 1] for k in range(2)]              # even / odd columns
                # tile q reads columns 2q .. 2q + 3: even[q], odd[q], even[q + 1], odd[q + 1]
                Dw = B1(cols[0][..., :nW], cols[1][..., :nW], cols[0][..., 1:nW + 1], cols[1][..., 1:nW + 1])
                for j in range(4):
                    # conv along H only (3 taps, padding 1): a conv2d with a (3, 1) kernel over [B, C, H, nW]
                    M[i][j] = mm(lambda a, c: F.conv2d(a, c.unsqueeze(-1), None, 1, (1, 0)), Dw[j], U2[i][j])
            Yt = [A1([M[i][j] for j in range(4)]) for i in range(4)]                    # along W: two columns per tile
            for col in range(2):
                y0, y1 = A1([Yt[i][col] for i in range(4)])                             # along T: two time steps
                out[:, :, 2 * p, :, col::2] = y0
                out[:, :, 2 * p + 1, :, col::2] = y1
        out = out[:, :, :T, :, :W]
    return out if b is None else out + b.view(1, -1, 1, 1, 1)


F.conv3d = conv_any
H, W, L = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 96, 16)))
torch.set_num_threads(8)
states = [("seed 0", lambda: synth.make_state_dict(0)), ("seed 1", lambda: synth.make_state_dict(1)), ("seed 2", lambda: synth.make_state_dict(2)),
          ("student-t4 gain 4", lambda: synth.make_state_dict(3, out_gain=4.0, tails="student"))]
rows = []
for name, make in states:
    sd = make()
    x = torch.from_numpy(OG.preprocess(synth.synthetic_frames(L + 1, H, W, seed=77))[None])
    clone = lambda dt=None: {k: (v.clone().double() if dt and v.is_floating_point() else v.clone()) for k, v in sd.items()}
    MODE["arm"] = "oracle"
    ref = U.forward(clone(), x)
    ref64 = U.forward(clone(True), x.double())
    e = lambda a, b: float((a.double() - b.double()).abs().max())
    line = {"state": name, "max_out": float(ref.max()), "oracle_f32_vs_f64": e(ref, ref64)}
    for arm in ("direct", "wino-T", "wino-TW"):
        MODE["arm"] = arm
        got = U.forward(clone(), x)
        line[arm] = {"vs_f32_oracle": e(got, ref), "vs_f64": e(got, ref64),
                     "worst_excess_over_bar": float(((got.double() - ref.double()).abs() - 1e-5 - 1e-5 * ref.double().abs()).max())}
    rows.append(line)
    print(line, flush=True)
import json
print(json.dumps({"H": H, "W": W, "L": L, "rows": rows}))
