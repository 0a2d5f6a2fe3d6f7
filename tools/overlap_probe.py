"""Developer tool: does SSIM overlap usefully with the compositing kernels when launched on a second
stream?  (independent launches, no data dependency; cfg3)"""
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
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
view.valid_count()
cam = view.cam
splats, radii, gb, ts, sg, _ = ops.project_bin_sort(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, model.budget)
rgb, dacc, fT, _ = ops.rasterize_fwd(cam, splats, sg, ts)
spec = model.loss_spec(view)
side = torch.cuda.Stream()
torch.cuda.synchronize()


def k7():
    ops.rasterize_bwd(cam, splats, gb, sg, ts, rgb, dacc, fT, loss=spec, want_tile_loss=True)


def k6():
    ops.rasterize_fwd(cam, splats, sg, ts)


def ssim():
    ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W))


def wall(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def both(a, b):
    def f():
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            b()
        a()
        ev2 = torch.cuda.Event(); ev2.record(side)
        torch.cuda.current_stream().wait_event(ev2)
    return f


print("k7 %.0f  k6 %.0f  ssim %.0f us" % (wall(k7), wall(k6), wall(ssim)))
print("k7 || ssim %.0f us   k6 || ssim %.0f us   k7 || k6 %.0f us" % (wall(both(k7, ssim)), wall(both(k6, ssim)), wall(both(k7, k6))))
