"""Developer soak: 160 steps at 300 k Gaussians / 720p with densification every 25 steps, SH ramp, the sync-free
budget started far too small (overflows, replays) and every next view announced (colour + front prefetch) against the
plain synchronous sequence -- Gaussian count, parameters and moments must come out bit-identical.
    python tools/soak_spec.py"""
import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd.densify import DensifyConfig
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dev = torch.device('cuda:0')
N, W, H, deg = 300_000, 1280, 720, 3
views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
def run(spec, announce):
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=20), params)
    m.enable_densification(DensifyConfig(warmup_length=20, refine_every=25, reset_alpha_every=3))
    if spec:
        m.enable_speculative_budget(capacity=400_000)      # far too small: overflows and replays
    for s in range(160):
        m.train_step(views[s % 8], next_view=views[(s + 1) % 8] if announce else None)
    m.flush(); torch.cuda.synchronize()
    return m
a = run(True, True); b = run(False, False)
print("N", a.params.N, b.params.N, "replays", getattr(a, "speculative_replays", 0), "steps", a.step, b.step)
print("params equal", a.params.N == b.params.N and torch.equal(a.params.flat, b.params.flat),
      "moments equal", a.params.N == b.params.N and torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq))
