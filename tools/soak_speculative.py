import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dev = torch.device('cuda:0')
N, W, H, deg = 50000, 640, 480, 3
views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=8) for v in range(8)]
P, _ = synthetic_gaussians(N, W, H, deg, 99)
def fresh():
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=100), params)
ref, spec = fresh(), fresh()
spec.enable_speculative_budget(capacity=1000, max_in_flight=6)
spec.budget.growth = 1.02          # grow as little as possible: many recoveries
t0 = time.time()
for step in range(1500):
    spec.train_step(views[(step * 3) % 8])
spec.flush(); torch.cuda.synchronize(); t1 = time.time()
for step in range(1500):
    ref.train_step(views[(step * 3) % 8])
torch.cuda.synchronize(); t2 = time.time()
print("replays", getattr(spec, "speculative_replays", 0), "capacity", spec.budget.capacity, "time spec %.2f s sync %.2f s" % (t1 - t0, t2 - t1))
print("identical:", torch.equal(spec.params.flat, ref.params.flat), torch.equal(spec.optimizer.exp_avg_sq, ref.optimizer.exp_avg_sq), spec.step, spec.optimizer.t)
