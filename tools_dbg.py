import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import torch_oracle as O
from tests.util import *
from touch_gs_amd import ops
dev=torch.device('cuda:0')
P, cam = scene(2000,160,96,3,1)
D = to_dev(P, dev)
sp = ops.project_fwd(amd_cam(cam), D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 3)
f = splat_fields(sp)
pr = O.project(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, 3)
both = pr['valid'] & (f['radius']==pr['radius']) & (f['radius']>0)
for k in ('xy','conic','rgb','depth'):
    e = relerr(f[k], pr[k].detach(), floor=1e-3)
    e = e.reshape(e.shape[0], -1).max(1)
    e[~both.numpy()] = 0
    i = int(e.argmax())
    print(k, e.max(), 'idx', i, 'hip', f[k][i].tolist(), 'ref', pr[k][i].tolist(), 'depth', pr['depth'][i].item(), 'radius', pr['radius'][i].item())
