"""What the two parts of a folded-tail launch differ by in scale: for res0-1 / dec0-2 conv2, log2 of
(x_scale * w_scale) / (t_scale * wt_scale) per sequence, on the bench's synthetic clip (GPU box)."""
import math, sys
import numpy as np, torch
sys.path.insert(0, ".")
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd.v2ce_3d import V2ce3d
from oracle import glue as OG

def p2(am):
    if not am > 0: return 1.0
    m, e = math.frexp(am); return 2.0 ** max(-100, min(100, 15 - e))

rows = []
orig = V2ce3d._conv
def conv(self, x0, x1, w_packed, *a, **k):
    tail = k.get("tail")
    y = orig(self, x0, x1, w_packed, *a, **k)
    if tail is not None:
        torch.cuda.synchronize()
        tx0, tx1, _, _, tw = tail
        wt = getattr(w_packed, "wt", False)
        wtail = w_packed[-8:].view(torch.float32).cpu().numpy() if wt else w_packed[-4:].view(torch.float32).cpu().numpy()
        ttail = tw[-4:].view(torch.float32).cpu().numpy()
        ax = x0.absmax[..., 0].reshape(-1).cpu().numpy()
        at = tx0.absmax[..., 0].reshape(-1).cpu().numpy()
        if tx1 is not None: at = np.maximum(at, tx1.absmax[..., 0].reshape(-1).cpu().numpy())
        for b in range(len(ax)):
            xs = p2(2 * ax[b]) if wt else p2(ax[b])
            rows.append((tuple(y.shape), b, float(ax[b]), float(at[b]), float(wtail[0]), float(ttail[0]),
                         math.log2(xs * wtail[1]) - math.log2(p2(at[b]) * ttail[1])))
    return y
V2ce3d._conv = conv
for seed in (0, 1, 3):
    sd = synth.make_state_dict(seed) if seed < 3 else synth.make_state_dict(3, out_gain=4.0, tails="student")
    m = V2ce3d(); m.load_state_dict(sd); m = m.eval().to("cuda")
    x = torch.from_numpy(np.stack([OG.preprocess(synth.synthetic_frames(17, 260, 346, seed=70 + s)) for s in range(2)])).cuda()
    rows.clear(); m(x)
    for r in rows: print(f"seed {seed} out {r[0]} seq {r[1]}: max|x| {r[2]:.3g} max|tail x| {r[3]:.3g} max|G| {r[4]:.3g} max|Wd'| {r[5]:.3g}  log2(main scale / tail scale) = {r[6]:+.0f}")
