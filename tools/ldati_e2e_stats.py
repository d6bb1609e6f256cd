#!/usr/bin/env python3
"""Debug aid (GPU box): plan and bucket occupancy of LDATI on real UNet output (bench.py's e2e step)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from v2ce_toolbox_amd import hip, synth                          # noqa: E402
from v2ce_toolbox_amd.LDATI import ldati_device                  # noqa: E402
from v2ce_toolbox_amd.v2ce_3d import V2ce3d                      # noqa: E402

m = V2ce3d()
m.load_state_dict(synth.make_state_dict(0))
m = m.eval().cuda()
x = bench.make_inputs(4, 0, "cuda")
vox = m(x).view(64, 2, 10, 260, 346)
print("voxel stats: max", float(vox.max()), "mean", float(vox.mean()), "frac>1", float((vox > 1).float().mean()),
      "frac==0", float((vox == 0).float().mean()))
ev = ldati_device(vox, fps=30, seed=1)
torch.cuda.synchronize()
y, keep, add, meta, tile_ws, ws = ev._keepalive
host = meta.cpu().numpy()
B = 64
max_n, max_tile, max_seg, total = (int(v) for v in host[B * 9 + 1:])
print("max_n", max_n, "max_tile", max_tile, "max_seg", max_seg, "total", total)
info = (ctypes.c_int64 * 10)()
hip.lib().v2ce_ldati_plan_info(B, 260, 346, 30.0, 0.0, None, total, max_seg, max_tile, info)
ok, shift, NB, T, capA, cap2, n_tab, n_bkt, lds_t, lds_s = list(info)
print("plan: ok", ok, "shift", shift, "NB", NB, "T", T, "capA", capA, "cap2", cap2, "lds", lds_t, lds_s)
w = ws.view(torch.int32).cpu().numpy()
o = 0
bofs = w[o:o + n_bkt].reshape(B * 9, NB + 1); o += n_bkt
groups = w[o:o + B * 9 * NB].reshape(B * 9, NB); o += 2 * B * 9 * NB
ngroups = w[o:o + B * 9]; o += B * 9
flag = w[o:o + B * 9]
btot = np.diff(bofs, axis=1)
print("big buckets:", int(w[o + B * 9 + 1]), " segments", B * 9, " bucket max", int(btot.max()), "mean", float(btot.mean()))
print("sort groups: total", int(ngroups.sum()), "per segment max", int(ngroups.max()), "mean", float(ngroups.mean()))
sizes = []
for sgm in range(B * 9):
    for g in range(int(ngroups[sgm])):
        b0, b1 = int(groups[sgm, g]) & 0xFFFF, int(groups[sgm, g]) >> 16
        sizes.append(int(bofs[sgm, b1] - bofs[sgm, b0]))
sizes = np.array(sizes)
print("group sizes: mean", float(sizes.mean()), "max", int(sizes.max()), "median", float(np.median(sizes)),
      "share of records in groups > 3072:", float(sizes[sizes > 3072].sum() / sizes.sum()))
