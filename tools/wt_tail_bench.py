"""Per-layer A/B on the GPU box: conv2 + folded shortcut of res0-1 / dec0-2 on the direct split-half kernel (v2ce_conv3d_fwd_tail)
vs the Winograd-T kernel (v2ce_conv3d_fwd_wt_tail).  B = 4 sequences x 16 frames of the 346x260 network.
python3 tools/wt_tail_bench.py [iters]"""
import sys
import torch

sys.path.insert(0, ".")
from v2ce_toolbox_amd import hip
from v2ce_toolbox_amd.v2ce_3d import V2ce3d

SHAPES = [  # name, C, H, W, tail C0 (upsampled unless C1 == 0), tail C1
    ("res.conv2", 512, 17, 22, 512, 0), ("dec0.conv2", 256, 33, 44, 512, 256), ("dec1.conv2", 128, 65, 87, 256, 128),
    ("dec2.conv2", 64, 130, 173, 128, 64),
]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T = 4, 16


def model():
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((8, 2), device="cuda")}
    return m


def act(C, H, W, g):
    Wp = V2ce3d._pitch(W)
    x = torch.relu(torch.randn(B, T, C // 16, H, Wp, 16, device="cuda", generator=g))
    x.lw, x.c16 = W, True
    x.absmax = x.abs().max().reshape(1)
    return x


for name, C, H, W, c0, c1 in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = act(C, H, W, g)
    if c1:
        tx0, tx1, up_to = act(c0, (H + 1) // 2, (W + 1) // 2, g), act(c1, H, W, g), (H, W)
    else:
        tx0, tx1, up_to = act(c0, H, W, g), None, None
    w = torch.randn(C, C, 3, 3, 3, device="cuda", generator=g) * (2.0 / (C * 27)) ** 0.5
    wd = torch.randn(C, c0 + c1, 1, 1, 1, device="cuda", generator=g) * (1.0 / (c0 + c1)) ** 0.5
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    m = model()
    twq = V2ce3d._pack(m, wd, split=True)
    out = {}
    for wt in (False, True):
        wq = V2ce3d._pack(m, w, None, V2ce3d._split_buffer(C, C, 27, "cuda", wt=wt), split=True)
        for with_tail in (False, True):
            ts = []
            for it in range(iters + 3):
                m._slot = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = V2ce3d._conv(m, x, None, wq, sc, sh, C, 3, 1, hip.ACT_RELU, split=True,
                                 tail=(tx0, tx1, up_to, 1, twq) if with_tail else None)
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            out[(wt, with_tail)] = (ts[len(ts) // 2], y)
    d = float((out[(True, True)][1][..., :W, :] - out[(False, True)][1][..., :W, :]).abs().max())
    print(f"{name:11s} C={C:3d} {H}x{W} tail {c0}+{c1}: direct {out[(False, False)][0]:.3f} -> +tail {out[(False, True)][0]:.3f} ms   "
          f"winograd-T {out[(True, False)][0]:.3f} -> +tail {out[(True, True)][0]:.3f} ms   max |d| {d:.2e}")
