import sys, time, torch, cProfile, pstats
sys.path.insert(0, '.')
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 100_000, 800, 800, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1235)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
views = [make_view(N, W, H, deg, 1235, dev, view=v, n_views=8) for v in range(8)]
for v in views: v.valid_count()
for i in range(20): model.train_step(views[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200): model.train_step(views[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("cpu enqueue ms/step", (t1 - t0) / 200 * 1e3, "total ms/step", (t2 - t0) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(200): model.train_step(views[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
