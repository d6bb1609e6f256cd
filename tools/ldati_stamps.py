#!/usr/bin/env python3
"""Phase shares of the LDATI tile pass / bucket sort from in-kernel s_memtime stamps (diagnostic
library `make -C v2ce-toolbox_amd/csrc libv2ce_hip_stamp.so`; shares only, never quote its run time).
    V2CE_HIP_LIB=v2ce-toolbox_amd/csrc/libv2ce_hip_stamp.so python tools/ldati_stamps.py [stress|sparse|e2e]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2ce_toolbox_amd import hip, synth                         # noqa: E402
from v2ce_toolbox_amd.LDATI import ldati_device                  # noqa: E402

regime = sys.argv[1] if len(sys.argv) > 1 else "stress"
if regime == "e2e":        # the voxels bench.py's e2e step hands to LDATI: V2ce3d on 4 synthetic sequences
    from oracle import glue as OG
    from v2ce_toolbox_amd.v2ce_3d import V2ce3d
    m = V2ce3d()
    m.load_state_dict(synth.make_state_dict(0), strict=True)
    m = m.eval().cuda()
    x = np.stack([OG.preprocess(synth.synthetic_frames(17, 260, 346, seed=1000 + s)) for s in range(4)])
    vox = m(torch.from_numpy(x).cuda()).view(64, 2, 10, 260, 346).contiguous()
    del m
else:
    vox = torch.from_numpy(synth.synthetic_voxels(24, 260, 346, seed=7, regime=regime)).cuda()
L = hip.lib()
buf = (ctypes.c_ulonglong * 32)()
for _ in range(2):
    ldati_device(vox, fps=30, seed=1)
torch.cuda.synchronize()
L.v2ce_debug_stamps(buf)
for _ in range(3):
    ldati_device(vox, fps=30, seed=1)
torch.cuda.synchronize()
L.v2ce_debug_stamps(buf)
v = np.array(buf[:], dtype=np.float64)
# ldati_tile_dense_kernel (round 4); V2CE_LDATI_OLD_TILE=1 stamps ldati_tile_pass_kernel with the same ten slots
# (loop head, P1 + scan, hist zero + P3 tables + barrier, P4 singles, P4 multis, barrier, P5 scan, P6 rank, barrier, P7 copy)
names_t = ["loop head / advance / loads", "D1 classify + scan + barrier A", "D2 work lists + barrier A2", "(unused)", "D3 timestamp batches",
           "barrier B", "D4 bucket scan + run table", "D5 rank", "barrier D", "copy-out"]
names_s = ["S0 setup", "S1 gather", "S2 widen+hist", "barrier", "S3 scan", "S4 rank", "barrier", "S5 emit"]
names_sp = ["1 loads + LDS clear + barrier", "2a relocate + classify + list append", "fused count: scans + barrier + reduce", "2b lists to registers + barrier",
            "3 singles", "3 multi-event pairs + barrier", "4 cell scan + barrier", "placement + barrier", "5 in-cell order + barrier", "6 runs + run table"]
if regime in ("e2e", "sparse"):
    names_t = names_sp                     # every tile takes ldati_tile_sparse_kernel<true> there (same ten slots)
for title, base, names in (("tile pass (wave 0 of every workgroup)", 0, names_t), ("bucket sort", 16, names_s)):
    tot = v[base:base + len(names)].sum()
    print(title)
    for i, n in enumerate(names):
        print(f"   {n:32s} {100 * v[base + i] / tot:6.1f} %   {v[base + i] / 3e6:9.2f} Mcycles per call (wave 0 of every workgroup)")
