"""Per-call timing of ldati_begin / finish on the stress chunk (which path each call takes: fused count or two-pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from v2ce_toolbox_amd import synth, LDATI
from v2ce_toolbox_amd.LDATI import ldati_begin
regime = sys.argv[1] if len(sys.argv) > 1 else "stress"
vox = torch.from_numpy(synth.synthetic_voxels(24, 260, 346, seed=7, regime=regime)).cuda()
pending = None
for k in range(8):
    prof = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q = ldati_begin(vox, fps=30, seed=1, profile=prof)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if pending is not None:
        pending.finish()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    pending = q
    print(k, "fused" if q.fused_ws is not None else "two-pass", f"begin {1e3*(t1-t0):.2f} ms (count events {prof[0][1].elapsed_time(prof[0][2]):.3f} ms), finish(prev) {1e3*(t2-t1):.2f} ms", LDATI._SEG_HINT)
