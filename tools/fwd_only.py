#!/usr/bin/env python3
"""V2ce3d forward only (no LDATI), timed with HIP events: for diagnostic library builds whose outputs are not valid
(V2CE_HIP_LIB=.../libv2ce_hip_ablate.so: epilogue-free upper bound)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
model = bench.fresh_model("f16x2", dev)
x = bench.make_inputs(4, 0, dev)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
if os.environ.get("ABLATE", "0") != "0":      # 1 = no epilogue, 2 = stores dropped by the range check, 3 = residual loads dropped
    # from here on the diagnostic library skips every conv epilogue; the activation buffers (same addresses: the
    # caching allocator replays the allocation sequence) and the absmax table keep the values of the last real forward,
    # so the MFMAs chew on realistic data (all-zero operands would run ~25 % faster on this power-limited chip)
    os.environ["V2CE_ABLATE_EPI"] = os.environ["ABLATE"]
    model._prep["absmax"].zero_ = lambda: None
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
n = int(os.environ.get("N", 8))
for _ in range(n):
    model(x)
t1.record()
torch.cuda.synchronize()
print(f"forward {t0.elapsed_time(t1) / n:.3f} ms per 64 frame-pairs ({os.environ.get('V2CE_HIP_LIB', 'product library')})")
