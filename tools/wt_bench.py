"""Per-layer A/B on the GPU box: the direct split-half kernel vs the Winograd-T kernel on the network's stride-1 3x3x3
shapes (B = 4 sequences x 16 frames, 346x260 input).  python3 tools/wt_bench.py [iters]"""
import sys
import torch

sys.path.insert(0, ".")
from v2ce_toolbox_amd import hip
from v2ce_toolbox_amd.v2ce_3d import V2ce3d

SHAPES = [  # name, C, H, W
    ("enc0/dec2.conv2", 64, 130, 173), ("enc1/dec1.conv2", 128, 65, 87), ("enc2/dec0.conv2", 256, 33, 44),
    ("enc3/res.conv", 512, 17, 22),
]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T = 4, 16


def model():
    m = V2ce3d.__new__(V2ce3d)
    torch.nn.Module.__init__(m)
    m._maps, m.precision, m._slot = {}, "f16x2", 0
    m._prep = {"absmax": torch.zeros((8, 2), device="cuda")}
    return m


for name, C, H, W in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    Wp = V2ce3d._pitch(W)
    x = torch.randn(B, T, C // 16, H, Wp, 16, device="cuda", generator=g)
    x.lw, x.c16 = W, True
    x.absmax = x.abs().max().reshape(1)
    res = torch.randn(B, T, C // 16, H, Wp, 16, device="cuda", generator=g)
    res.lw, res.c16 = W, True
    w = torch.randn(C, C, 3, 3, 3, device="cuda", generator=g) * (2.0 / (C * 27)) ** 0.5
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    m = model()
    out = {}
    for wt in (False, True):
        buf = V2ce3d._split_buffer(C, C, 27, "cuda", wt=wt)
        wq = V2ce3d._pack(m, w, None, buf, split=True)
        for with_res in (False, True):
            ts = []
            for it in range(iters + 3):
                m._slot = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                y = V2ce3d._conv(m, x, None, wq, sc, sh, C, 3, 1, hip.ACT_RELU, residual=res if with_res else None, split=True)
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            out[(wt, with_res)] = (ts[len(ts) // 2], y)
    flops = 2.0 * B * T * H * W * C * C * 27
    d = float((out[(True, True)][1][..., :W, :] - out[(False, True)][1][..., :W, :]).abs().max())
    print(f"{name:18s} C={C:3d} {H}x{W}: direct {out[(False, False)][0]:.3f} / +res {out[(False, True)][0]:.3f} ms "
          f"({flops / out[(False, True)][0] / 1e9:.0f} TF)   winograd-T {out[(True, False)][0]:.3f} / +res {out[(True, True)][0]:.3f} ms "
          f"({flops / out[(True, True)][0] / 1e9:.0f} TF-equivalent)   max |d| {d:.2e}")
