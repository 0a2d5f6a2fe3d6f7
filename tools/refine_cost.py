"""Developer tool: where the time of a refinement step goes (cfg3, densifying trainer settings of bench.py's
train_densify run).  Prints wall ms (device-synchronised) of the steps around each refinement."""
import sys, time, torch
sys.path.insert(0, ".")
from touch_gs_amd.densify import DensifyConfig
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device("cuda:0")
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
cfg = ModelConfig(sh_degree=deg, sh_degree_interval=100, spatial_sort=True, resort_every_refines=4)
m = DepthGaussianSplattingModel(cfg, params)
m.spatial_sort()
m.enable_densification(DensifyConfig(warmup_length=100, refine_every=100, reset_alpha_every=3, max_gaussians=int(1.6 * N)))
m.enable_speculative_budget()
views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
for v in views:
    v.valid_count()
sync = torch.cuda.synchronize
import touch_gs_amd.densify as D
orig_refine = D.DensityController.refine
def timed_refine(self, *a, **k):
    sync(); t0 = time.perf_counter()
    out = orig_refine(self, *a, **k)
    sync(); print(f"    refine() itself {1e3 * (time.perf_counter() - t0):7.2f} ms  {out[2]}")
    return out
D.DensityController.refine = timed_refine
orig_sort = m.spatial_sort
def timed_sort():
    sync(); t0 = time.perf_counter(); r = orig_sort(); sync()
    print(f"    spatial_sort    {1e3 * (time.perf_counter() - t0):7.2f} ms"); return r
m.spatial_sort = timed_sort
for i in range(520):
    near = (i + 1) % 100 == 0 or i % 100 in (0, 1, 2)
    if near:
        sync(); t0 = time.perf_counter()
    m.train_step(views[i % 8], next_view=views[(i + 1) % 8])
    if near:
        sync(); print(f"step {i + 1:4d}: {1e3 * (time.perf_counter() - t0):8.2f} ms   N = {m.params.N}  replays {getattr(m, 'speculative_replays', 0)}")
