import sys, torch, numpy as np
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
p = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
b = ops.IntersectBudget()
splats, radii, gb, ts, sg, st = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, b)
np.save('gpurun_out/tile_start.npy', ts.cpu().numpy())
