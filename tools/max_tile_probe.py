import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from v2ce_toolbox_amd import synth
from v2ce_toolbox_amd.LDATI import ldati_begin
vox = torch.from_numpy(synth.synthetic_voxels(24,260,346,seed=7,regime="stress")).cuda()
q = ldati_begin(vox, fps=30, seed=1)
q.ready.synchronize()
h = q.host.numpy()
print("max_n,max_tile,max_seg,total", h[24*9+1:])
