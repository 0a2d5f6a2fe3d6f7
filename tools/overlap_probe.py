"""Developer probe: do the compositing kernels (VALU-bound) and the SSIM kernels (latency-bound) overlap
when they run on two streams?  Prints K6 alone, SSIM alone, both concurrently; same for K7."""
import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
model.spatial_sort()
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
p = model.params
sp, _, gb, ts, sg, st = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
rgb, dacc, fT, _ = ops.rasterize_fwd(view.cam, sp, sg, ts)
_, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W), reduce=False)
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()

def k6(): ops.rasterize_fwd(view.cam, sp, sg, ts)
def k7(): ops.rasterize_bwd(view.cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=v_img, loss=model.loss_spec(view), want_tile_loss=True)
def ssim(): ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W), reduce=False)

def timed(fa, fb, reps=10):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if fb is not None:
            s2.wait_event(e0)
            with torch.cuda.stream(s2):
                fb()
                eb = torch.cuda.Event(); eb.record(s2)
        if fa is not None:
            fa()
        if fb is not None:
            main.wait_event(eb)
        e1.record(main)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[len(out) // 2]

for name, f in (("K6", k6), ("K7", k7)):
    a, b, ab = timed(f, None), timed(None, ssim), timed(f, ssim)
    print(f"{name} alone {a:.0f} us, SSIM alone (side stream) {b:.0f} us, concurrent {ab:.0f} us  (sum {a + b:.0f})")
