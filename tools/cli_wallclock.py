#!/usr/bin/env python3
"""CLI wall-clock INCLUDING the events file (VERDICT r2 missing #3): a 2048-frame 346x260 clip (BASELINE config 3's clip on
one GPU, -b 32) through v2ce.run, (a) streamed into the .npz while the clip runs, (b) whole clip in host memory, then one
np.savez like the reference (v2ce.py:363-372).  Prints one JSON line.   python tools/cli_wallclock.py [frames] [out_dir]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2ce_toolbox_amd import synth                       # noqa: E402
from v2ce_toolbox_amd import v2ce as cli                 # noqa: E402
from v2ce_toolbox_amd.v2ce_3d import V2ce3d              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out_dir = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
frames = synth.synthetic_frames(n, 260, 346)


def model():
    m = V2ce3d()
    m.load_state_dict(synth.make_state_dict(0))
    return m.eval().to("cuda")


kw = dict(infer_type="center", batch_size=32, fps=30, seed=11)
cli.run(frames[:513], model(), **kw)                     # warm-up: allocators, kernels
res = {}
for mode in ("streamed", "savez_at_end", "streamed", "savez_at_end"):
    path = os.path.join(out_dir, f"cli_wallclock_{mode}.npz")
    m = model()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "streamed":
        count = cli.run(frames, m, out_path=path, **kw)
    else:
        ev = cli.run(frames, m, **kw)
        t1 = time.perf_counter()
        np.savez(path, event_stream=ev)
        count = len(ev)
        res["savez_alone_s"] = time.perf_counter() - t1
        del ev
    dt = time.perf_counter() - t0
    res[mode] = {"wall_s": dt, "frame_pairs_per_s": (n - 1) / dt, "file_GB": os.path.getsize(path) / 1e9, "events": int(count)}
    os.remove(path)
print(json.dumps({"frames": n, "batch": 32, "out_dir": out_dir, **res}))
