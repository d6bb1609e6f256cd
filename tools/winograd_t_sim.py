"""Numerics experiment (CPU, oracle-side only): what a Winograd F(2,3) transform along T would cost the parity bar.
Every stride-1 3x3x3 convolution of the oracle UNet is replaced by its F(2,3)-along-T form in f32 (input transform,
transformed weights, products and accumulation in f32, output transform in f32) and the network output compared with the
direct oracle and with an f64 run.  Not product code."""
import sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import unet as U
from oracle import glue as OG
from v2ce_toolbox_amd import synth

direct = F.conv3d
MODE = {"wino": False}


def wino_t(x, w, b, stride, pad):
    s = stride if isinstance(stride, int) else None
    if not MODE["wino"] or w.shape[2:] != (3, 3, 3) or (s != 1 and tuple(stride) != (1, 1, 1)):
        return direct(x, w, b, stride, pad)
    # x [B,C,T,H,W]; pad T by 1 both sides (+1 more when T is odd)
    B, C, T, H, W = x.shape
    Tp = T + (T & 1)
    xp = F.pad(x, (0, 0, 0, 0, 1, 1 + Tp - T))
    g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
    G = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    out = x.new_zeros(B, w.shape[0], Tp, H, W)
    for p in range(Tp // 2):
        d0, d1, d2, d3 = (xp[:, :, 2 * p + i] for i in range(4))
        D = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        m = [F.conv2d(D[i], G[i], None, 1, 1) for i in range(4)]
        out[:, :, 2 * p] = m[0] + m[1] + m[2]
        out[:, :, 2 * p + 1] = m[1] - m[2] - m[3]
    out = out[:, :, :T]
    if b is not None:
        out = out + b.view(1, -1, 1, 1, 1)
    return out


F.conv3d = wino_t
H, W, L = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 96, 16)))
torch.set_num_threads(8)
for seed in (0, 1):
    sd = synth.make_state_dict(seed)
    x = torch.from_numpy(OG.preprocess(synth.synthetic_frames(L + 1, H, W, seed=77))[None])
    MODE["wino"] = False
    ref = U.forward({k: v.clone() for k, v in sd.items()}, x)
    ref64 = U.forward({k: v.clone().double() if v.is_floating_point() else v.clone() for k, v in sd.items()}, x.double())
    MODE["wino"] = True
    got = U.forward({k: v.clone() for k, v in sd.items()}, x)
    e = lambda a, b: float(((a.double() - b.double()).abs() - 1e-5 * b.double().abs()).max())
    print(f"seed {seed}: max {float(ref.max()):.2f}; direct f32 vs f64 {e(ref, ref64):.2e}; winograd-T f32 vs f64 {e(got, ref64):.2e}; "
          f"winograd-T vs direct f32 (the parity bar, 1e-5) {e(got, ref):.2e}")
