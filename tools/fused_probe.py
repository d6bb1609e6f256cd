import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd.LDATI import ldati_begin
vox = torch.from_numpy(synth.synthetic_voxels(3, 260, 346, seed=12, regime="sparse")).cuda() * 0.2
q = ldati_begin(vox, fps=30, seed=1)
q.ready.synchronize()
print("stats", q.host.numpy()[3 * 9 + 1:])
ev = q.finish()
torch.cuda.synchronize()
print(ev.num_events)
