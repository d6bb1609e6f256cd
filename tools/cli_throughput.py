#!/usr/bin/env python3
"""PCIe-inclusive throughput of the drop-in driver (v2ce.run): uint8 frames in host memory ->
event_stream structured array in host memory, 346x260, --batch_size 4."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd import v2ce as cli
from v2ce_toolbox_amd.v2ce_3d import V2ce3d

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = np.concatenate([synth.synthetic_frames(17, 260, 346, seed=1000 + s)[:16] for s in range(n_seq)] +
                        [synth.synthetic_frames(1, 260, 346, seed=1)])
m = V2ce3d(); m.load_state_dict(synth.make_state_dict(0)); m = m.eval().to("cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev = cli.run(frames, m, batch_size=4, stage2_batch_size=int(os.environ.get("S2B", 24)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{len(frames)-1} pairs, {len(ev)/1e6:.1f} M events: {dt*1e3:.1f} ms -> {(len(frames)-1)/dt:.1f} frame-pairs/s, {len(ev)/dt/1e6:.1f} Mev/s")
