"""Developer tool: train-step time, eager launches vs hipGraph replay (cfg2 and cfg3)."""
import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
cfgs = {"cfg2": (100_000, 800, 800, 3, 1235, 400), "cfg3": (1_000_000, 1920, 1080, 3, 1236, 100)}
dev = torch.device('cuda:0')
for name, (N, W, H, deg, seed, n) in cfgs.items():
    P, _ = synthetic_gaussians(N, W, H, deg, seed)
    views = [make_view(N, W, H, deg, seed, dev, view=v, n_views=8) for v in range(8)]
    for mode in ("eager", "graph"):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
        if mode == "graph":
            model.capture_step_graphs(views)
        else:
            model.train_step(views[0])
            model.budget = ops.IntersectBudget(capacity=int(model.budget.capacity * 1.3), sync=False)
        for i in range(40): model.train_step(views[i % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n): model.train_step(views[i % 8])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        model.budget.check()
        print(name, mode, "ms/step %.4f" % ((t2 - t0) / n * 1e3), "(host enqueue %.4f)" % ((t1 - t0) / n * 1e3))
        del model, params
        torch.cuda.empty_cache()
