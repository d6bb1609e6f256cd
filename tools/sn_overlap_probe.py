#!/usr/bin/env python3
"""Upper bound of what the spectral-norm side stream costs the convolutions it overlaps: time the forward
pass with the per-call power iteration + weight re-pack enabled, and with it skipped (weights of the
first call reused).  Diagnostic only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import glue as OG                                   # noqa: E402
from v2ce_toolbox_amd import synth                               # noqa: E402
from v2ce_toolbox_amd.v2ce_3d import V2ce3d                      # noqa: E402

m = V2ce3d()
m.load_state_dict(synth.make_state_dict(0), strict=True)
m = m.eval().cuda()
x = torch.from_numpy(np.stack([OG.preprocess(synth.synthetic_frames(17, 260, 346, seed=1000 + s)) for s in range(4)])).cuda()


def timed(n=8):
    m(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        m(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


a = timed()
V2ce3d._launch_sn = lambda self: None
b = timed()
print(f"forward with SN stream {a:.3f} ms, without {b:.3f} ms")

# the spectral-norm work alone (12 power iterations + re-packs, nothing else on the GPU)
m2 = V2ce3d()
m2.load_state_dict(synth.make_state_dict(0), strict=True)
m2 = m2.eval().cuda()
m2.advance_spectral_norm()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    m2.advance_spectral_norm()
torch.cuda.synchronize()
print(f"spectral norm + re-pack alone: {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per call")
