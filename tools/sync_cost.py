"""Developer tool: train-step time with the synchronous intersection budget against the pre-sized
sync-free budget bench.py uses and the speculative sync-free budget the trainer uses."""
import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
cfgs = {"cfg2": (100_000, 800, 800, 3, 1235), "cfg3": (1_000_000, 1920, 1080, 3, 1236)}
for name, (N, W, H, deg, seed) in cfgs.items():
    dev = torch.device('cuda:0')
    P, _ = synthetic_gaussians(N, W, H, deg, seed)
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    views = [make_view(N, W, H, deg, seed, dev, view=v, n_views=8) for v in range(8)]
    for v in views: v.valid_count()
    for mode in ("sync", "presized", "speculative"):
        if mode == "presized":
            model.budget = ops.IntersectBudget(capacity=int(model.budget.capacity * 1.25), sync=False)
        if mode == "speculative":
            model.enable_speculative_budget()
        for i in range(16): model.train_step(views[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200
        for i in range(n): model.train_step(views[i % 8])
        model.flush()
        torch.cuda.synchronize()
        print(name, mode, "ms/step", round((time.perf_counter() - t0) / n * 1e3, 4))
    del model, params, views; torch.cuda.empty_cache()
