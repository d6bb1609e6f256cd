#!/usr/bin/env python3
"""In-network sweep of conv output boxes: one bench.py run per candidate (V2CE_BOX_<Ho>x<Wo>_<s>_<pos>),
reports ms per step.  Isolated-layer timings (tools/tile_probe.py) do not transfer to the network
(same-kernel loops run from L2 and at another power state), so boxes are chosen here.
    python tools/box_sweep.py 260x346_1_512 4,4,32 8,4,16 4,8,16"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key = sys.argv[1]
for cand in sys.argv[2:]:
    env = dict(os.environ)
    env["V2CE_BOX_" + key] = cand
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                          "--no-exact-f32", "--no-host-to-host"], env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(f"{key} {cand:10s} {d['ms_per_step']:.3f} ms/step", flush=True)
