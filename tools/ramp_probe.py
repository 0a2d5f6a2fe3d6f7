"""Developer tool: train-step time as a function of the step index (clock / cache warm-up), cfg3."""
import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
p = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), p)
views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
for v in views: v.valid_count()
model.train_step(views[0])
model.budget = ops.IntersectBudget(capacity=int(model.budget.capacity * 1.3), sync=False)
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)   # idle gap before the run
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for blk in range(40):
    for i in range(10):
        model.train_step(views[i % 8])
    evs[blk + 1].record()
torch.cuda.synchronize()
print("ms/step per block of 10 steps:", " ".join("%.3f" % (evs[i].elapsed_time(evs[i + 1]) / 10) for i in range(40)))
