"""Developer tool: the front half in its unfused form (K1, then count / scan / fill / sort) next to the
fused call, cfg3 in Morton layout, for rocprofv3 kernel traces."""
import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
model.spatial_sort()
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
p = model.params
b = ops.IntersectBudget()
for _ in range(6):
    sp = ops.project_fwd(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
    ops.bin_sort(view.cam, sp, b)
    ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, b)
torch.cuda.synchronize()
