#!/usr/bin/env python3
"""Time the ablation samplers (csrc/sampler.hip) on a C5-sized chunk: count + emit + sort + unpack, HIP events."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from v2ce_toolbox_amd import hip, sample_methods as SM, synth  # noqa: E402

for regime, B in (("stress", 24), ("sparse", 24)):
    y = torch.from_numpy(synth.synthetic_voxels(B, 260, 346, seed=1, regime=regime)).cuda()
    for name, mode in (("random", hip.SAMPLER_RANDOM), ("even", hip.SAMPLER_EVEN), ("pure_slope", hip.SAMPLER_PURE_SLOPE)):
        for _ in range(2):
            ev = SM.sampler_device(y, mode, 0, 30, seed=3)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            ev = SM.sampler_device(y, mode, 0, 30, seed=3)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        n = ev.num_events
        print(f"{regime:7s} {name:10s} {n / 1e6:7.1f} Mevents  {ms:7.2f} ms  {n / ms / 1e3:8.1f} Mevents/s  "
              f"{(y.numel() * 4 + 13 * n) / ms / 1e6:7.1f} GB/s algorithmic")
