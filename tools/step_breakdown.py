#!/usr/bin/env python3
"""Where does a bench step go?  Times the stage-1 forward and the stage-2 calls separately with
HIP events and host clocks (b = 4 sequences, 346x260)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd.LDATI import ldati_device
from v2ce_toolbox_amd.v2ce_3d import V2ce3d

dev = torch.device("cuda:0")
m = V2ce3d(precision=os.environ.get("PRECISION", "f16x2")); m.load_state_dict(synth.make_state_dict(0)); m = m.eval().to(dev)
x = bench.make_inputs(4, 0, dev)
for _ in range(2):
    vox = m(x)
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(4)]
    t0 = time.perf_counter(); e[0].record()
    vox = m(x)
    e[1].record(); t1 = time.perf_counter()
    d = ldati_device(vox.view(64, 2, 10, 260, 346), fps=30, seed=1)
    e[2].record(); t2 = time.perf_counter()
    pk = d.packed()
    e[3].record(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"forward gpu {e[0].elapsed_time(e[1]):7.2f} ms (host launch {1e3*(t1-t0):6.2f}) | ldati gpu {e[1].elapsed_time(e[2]):6.2f} ms (host {1e3*(t2-t1):6.2f}) | pack {e[2].elapsed_time(e[3]):5.2f} | wall {1e3*(t3-t0):7.2f} ms")
