"""Developer tool: a few full train steps on cfg3 (fused or separate Adam) for rocprofv3 passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
model.fuse_adam = (len(sys.argv) < 3 or sys.argv[2] != 'separate')
model.spatial_sort()   # the framework's default layout (trainer, bench.py)
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
view.valid_count()
prefetch = len(sys.argv) > 2 and sys.argv[2] == 'prefetch'   # colour prefetch for the (same) next view
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    model.train_step(view, next_view=view if prefetch else None)
torch.cuda.synchronize()
