"""Developer tool: K6/K7 time with blocks visiting tiles longest-list-first (per XCD band / globally)
instead of in spatial order (the schedule tgs_bin_sort emits as tile_order)."""
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
order = ts.tile_order


def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


ref = ops.rasterize_fwd(cam, splats, sg, ts)[0]
for rnd in range(3):            # alternate: GPU clocks / caches drift during the first launches
    for name in ("longest-first per XCD band (product)", "spatial"):
        if name == "spatial":
            del ts.tile_order          # the wrappers then pass NULL: spatial order
        else:
            ts.tile_order = order
        k6 = t(lambda: ops.rasterize_fwd(cam, splats, sg, ts))
        k7 = t(lambda: ops.rasterize_bwd(cam, splats, gb, sg, ts, rgb, dacc, fT, loss=spec, want_tile_loss=True))
        out = ops.rasterize_fwd(cam, splats, sg, ts)[0]
        print("%-38s K6 %.1f us  K7 %.1f us  same image: %s" % (name, k6, k7, bool(torch.equal(ref, out))))
ts.tile_order = order
n = (ts[1:] - ts[:-1])
o = order.long()
T = cam.num_tiles
ok = sorted(o[o < T].tolist()) == list(range(T))
print("order is a permutation of the tiles:", ok)
