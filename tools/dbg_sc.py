import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch.nn.functional as F
from v2ce_toolbox_amd import hip
from v2ce_toolbox_amd.v2ce_3d import V2ce3d
cin, cout, s, H, W = 32, 32, 1, 8, 32
g = torch.Generator().manual_seed(1)
x = torch.randn(1, cin, 3, H, W, generator=g)
w = torch.zeros(cout, cin, 3, 3, 3)
wd = torch.zeros(cout, cin, 1, 1, 1)
for o in range(cout):
    wd[o, o % cin] = 1.0 + o          # shortcut output o = (1 + o) * x[o % cin]
w[:, :, 1, 1, 1] = wd[:, :, 0, 0, 0]
m = V2ce3d.__new__(V2ce3d); torch.nn.Module.__init__(m)
m._maps, m.precision, m._slot = {}, "f16x2", 0
m._prep = {"absmax": torch.zeros((4, 2), device="cuda")}
xd = x.permute(0, 2, 1, 3, 4).contiguous().cuda(); xd.absmax = xd.abs().max().reshape(1)
one, zero = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
try:
    y, ysc = V2ce3d._conv(m, V2ce3d.to_c16(xd), None, V2ce3d._pack(m, w.cuda().contiguous(), split=True), one, zero, cout, 3, s, hip.ACT_NONE,
                          split=True, dense_out=True, sc=(V2ce3d._pack(m, wd.cuda().contiguous(), split=True), one, zero))
except Exception as e:
    print("err", e); raise
got = V2ce3d.to_planar(ysc).permute(0, 2, 1, 3, 4).cpu().numpy()[0]      # [C, T, H, W]
gy = V2ce3d.to_planar(y).permute(0, 2, 1, 3, 4).cpu().numpy()[0]
print('max |y - ysc| per k:', [float(np.abs(gy[k::4] - got[k::4]).max()) for k in range(4)])
print('y vs expected per k:', [float(np.abs(gy[k::4] - np.stack([(1.0 + o) * xs_ for o, xs_ in zip(range(k, cout, 4), x.numpy()[0][k::4])])).max()) for k in range(4)])
xs = x.numpy()[0]
for o in (0, 1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 31):
    # find which input channel / scale the output equals
    best = None
    for c in range(cin):
        r = got[o] / np.where(np.abs(xs[c]) > 1e-3, xs[c], np.nan)
        med = np.nanmedian(r)
        if np.nanmax(np.abs(r - med)) < 1e-3 * abs(med) + 1e-3:
            best = (c, med)
    print("out", o, "= input", best, "expected", (o % cin, 1.0 + o))
print("---- k=1 channels: closest single input channel")
for o in (1, 5, 9, 13, 17, 21):
    best = (1e9, None, None)
    for c in range(cin):
        den = np.where(np.abs(xs[c]) > 0.05, xs[c], np.nan)
        r = got[o] / den
        med = np.nanmedian(r)
        dev = np.nanmedian(np.abs(r - med))
        if dev < best[0]:
            best = (dev, c, med)
    print("out", o, "closest input", best[1], "scale", best[2], "dev", best[0], " sample got", got[o].ravel()[:3], "x[o]", xs[o % cin].ravel()[:3])
print("---- where is channel 1 wrong")
exp = 2.0 * xs[1]
bad = np.argwhere(np.abs(got[1] - exp) > 1e-3)
print(len(bad), "of", exp.size, "positions wrong; first:", bad[:12].tolist())
print("values got/exp:", [(float(got[1][tuple(b)]), float(exp[tuple(b)])) for b in bad[:4]])
for b in bad[:3]:
    t, h, w = b
    # is it another channel's value at the same position?
    cands = [(c, (1.0 + c) * xs[c][t, h, w]) for c in range(cin)]
    print(tuple(b), "got", got[1][t, h, w], "matches channel", [c for c, v in cands if abs(v - got[1][t, h, w]) < 1e-4])
