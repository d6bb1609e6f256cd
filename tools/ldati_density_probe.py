#!/usr/bin/env python3
"""Tile-pass time per event against density and workgroup shape (does a second workgroup per CU pay?).
    python tools/ldati_density_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2ce_toolbox_amd.LDATI import ldati_device                  # noqa: E402

rng = np.random.default_rng(3)
for scale in (6.0, 3.0, 1.5):
    vox = torch.from_numpy((scale * rng.random((24, 2, 10, 260, 346))).astype(np.float32)).cuda()
    for threads in (os.environ.get("V2CE_LDATI_TILE_THREADS", "auto"),):     # read once by the library: one process per setting
        for _ in range(2):
            ev = ldati_device(vox, fps=30, seed=1)
        torch.cuda.synchronize()
        prof = []
        for _ in range(5):
            ev = ldati_device(vox, fps=30, seed=1, profile=prof)
        torch.cuda.synchronize()
        ms = sum(e0.elapsed_time(e1) for name, e0, e1, _ in prof if name == "emit") / 5
        print(f"scale {scale}: {ev.num_events / 1e6:7.1f} M events, tile threads {threads}: emit {ms:.3f} ms = {ms * 1e6 / ev.num_events:.2f} ps... ns/kev {ms * 1e6 / (ev.num_events / 1e3):.1f}")
