import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dev = torch.device('cuda:0')
N, W, H, deg = 1_000_000, 1920, 1080, 3
views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
P, _ = synthetic_gaussians(N, W, H, deg, 4321)     # a different scene: real optimisation, I drifts a lot
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=200), params)
model.enable_speculative_budget()
t0 = time.time()
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    v = views[step % 8]
    model.train_step(v)
    if (step + 1) % 500 == 0:
        model.flush()
        loss = model.loss_from(model.last["tile_loss"], model.last["ssim_sum"], v)
        print(step + 1, {k: round(float(x), 5) for k, x in loss.items()}, "n_isect", int(model.last["status"][0]),
              "cap", model.budget.capacity, "replays", getattr(model, "speculative_replays", 0),
              "%.1f it/s" % ((step + 1) / (time.time() - t0)), flush=True)
model.flush()
print("finite:", bool(torch.isfinite(model.params.flat).all()), "step", model.step, "t", model.optimizer.t)
