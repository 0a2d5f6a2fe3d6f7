"""Developer tool: K6's split factor and K7's four-wave rule on the saved 720p checkpoints (same process, alternating).
    python tools/k6k7_rules.py [bunny|block]"""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from touch_gs_amd import ops
import ckpt_loop
what = sys.argv[1] if len(sys.argv) > 1 else "bunny"
dev = torch.device("cuda:0")
if what in ("cfg3", "clustered"):
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    from touch_gs_amd.optim import GaussianParams, morton_order
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    N, W, H = 1_000_000, 1920, 1080
    P, _ = synthetic_gaussians(N, W, H, 3, 1236, clustered=what == "clustered")
    perm = morton_order(P["means"])
    params = GaussianParams.from_tensors(*[P[k][perm].to(dev) for k in GaussianParams.NAMES])
    m = DepthGaussianSplattingModel(ModelConfig(sh_degree=3, sh_degree_interval=0), params)
    views = [make_view(N, W, H, 3, 1236, dev, view=v, n_views=8, clustered=what == "clustered") for v in range(0, 8, 2)]
    for v in views: v.valid_count()
else:
    m, views = ckpt_loop.load(os.path.join(ROOT, "build_ab/ckpt/model_%s_1.pt" % ("bunny_real" if what == "bunny" else "block")), dev, n_views=8)
p, deg = m.params, 3
k6 = {f"split{f}": ops.raster_opts(k6_split=f) for f in (4, 3, 2)}
k7 = {f"quad{f}_min{w}": ops.raster_opts(k7_quad=f, k7_quad_min_walk=w) for f, w in ((8, 48), (8, 32), (8, 24), (8, 16), (8, 12), (8, 8), (8, 4))}
r6, r7 = {k: [] for k in k6}, {k: [] for k in k7}
def t(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); out = fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3, out
for v in views:
    cam = v.cam; H, W = cam.H, cam.W
    sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
    per6, per7 = {k: [] for k in k6}, {k: [] for k in k7}
    for rep in range(6):
        for k, o in k6.items():
            us, out = t(lambda: ops.rasterize_fwd(cam, sp, sg, ts, opts=o))
            if rep: per6[k].append(us)
    rgb, dacc, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    ss, vimg = ops.ssim_fwd_bwd(rgb, v.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    loss = m.loss_spec(v)
    for rep in range(4):
        for k, o in k7.items():
            # the walk statistics K7's rule reads ACCUMULATE over forwards on the same lists: fresh lists + ONE forward per backward
            sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
            rgb, dacc, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
            torch.cuda.synchronize()
            us, _ = t(lambda: ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=loss, want_tile_loss=True, opts=o))
            if rep: per7[k].append(us)
    for k in k6: r6[k].append(sorted(per6[k])[2])
    for k in k7: r7[k].append(sorted(per7[k])[1])
mean = lambda d: {k: round(sum(v) / len(v), 1) for k, v in d.items()}
print(json.dumps({"workload": what, "k6_mean_us": mean(r6), "k7_mean_us": mean(r7)}))
