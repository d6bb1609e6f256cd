#!/usr/bin/env python3
"""How long the HOST takes to enqueue one step (V2ce3d forward + LDATI count) versus how long the GPU takes to run it:
if the first approaches the second, a slower host CPU paces the pipeline (host_to_host box dependence)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from v2ce_toolbox_amd import LDATI  # noqa: E402

dev = torch.device("cuda:0")
model = bench.fresh_model("f16x2", dev)
x = bench.make_inputs(4, 0, dev)
for _ in range(3):
    y = model(x)
torch.cuda.synchronize()
host, n = 0.0, 10
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter()
    y = model(x)
    vox = y.reshape(-1, 2, 10, y.shape[-2], y.shape[-1])
    pend = LDATI.ldati_begin(vox, 0, 30, seed=1)
    host += time.perf_counter() - t0
    ev = pend.finish()
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print(f"host enqueue {1e3 * host / n:.2f} ms per step; wall {1e3 * total / n:.2f} ms per step "
      f"(cpu: {os.cpu_count()} threads)")
import cProfile, pstats  # noqa: E402
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    y = model(x)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
