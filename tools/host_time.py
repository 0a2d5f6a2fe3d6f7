"""Developer tool: host (Python + ctypes) time per train step against the GPU time (cfg3)."""
import sys, time, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
for pipe in (False, True):
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0, pipeline_ssim=pipe), params)
    m.spatial_sort()
    b = ops.IntersectBudget()
    ops.project_bin_sort(views[0].cam, params.means, params.log_scales, params.quats, params.opac_logit, params.sh, deg, b)
    m.budget = ops.IntersectBudget(capacity=int(b.last_need * 1.3) + 4096, sync=False)
    for i in range(30):
        m.train_step(views[i % 8], next_view=views[(i + 1) % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for i in range(n):
        m.train_step(views[i % 8], next_view=views[(i + 1) % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"pipeline_ssim={pipe}: host {1e3 * (t1 - t0) / n:.3f} ms/step enqueue, {1e3 * (t2 - t0) / n:.3f} ms/step total", flush=True)
