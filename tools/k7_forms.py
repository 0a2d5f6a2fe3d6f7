"""Developer tool: K7 in its forms on the same frames, alternating in one process (same box, same clocks).
    python tools/k7_forms.py [cfg3|bunny] [reps]   ->  us per view and 8-view mean for one wave / 4x4 blocks (/ quad rule)"""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from touch_gs_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
if what == "cfg3":
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    from touch_gs_amd.optim import GaussianParams, morton_order
    N, W, H, deg = 1_000_000, 1920, 1080, 3
    P, _ = synthetic_gaussians(N, W, H, deg, 1236)
    perm = morton_order(P["means"])
    D = {k: v[perm].to(dev).contiguous() for k, v in P.items()}
    views = [make_view(N, W, H, deg, 1236, dev, view=v, n_views=8) for v in range(8)]
else:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ckpt_loop
    m, views = ckpt_loop.load(os.path.join(ROOT, "build_ab/ckpt/model_bunny_real_1.pt"), dev)
    p = m.params
    D, deg = dict(means=p.means, log_scales=p.log_scales, quats=p.quats, opac_logit=p.opac_logit, sh=p.sh), 3
forms = {"one_wave": ops.raster_opts(k7_quad=0, k7_blocks=0), "blocks": ops.raster_opts(k7_blocks=1),
         "default_rule": ops.raster_opts(k7_blocks=0)}
res = {k: [] for k in forms}
for v in views:
    cam = v.cam
    H, W = cam.H, cam.W
    sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
    rgb, dacc, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    ss, vimg = ops.ssim_fwd_bwd(rgb, v.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    loss = dict(gt_rgb=v.rgb, l1_weight=0.8 / (3 * H * W), gt_depth=v.depth, depth_weight=0.2 / max(v.valid_count(), 1),
                uncertainty=v.uncertainty, uncertainty_weight=1.0, eps=1e-6)
    per = {k: [] for k in forms}
    for r in range(reps + 1):
        for name, o in forms.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=loss, want_tile_loss=True, opts=o)
            b.record()
            torch.cuda.synchronize()
            if r:
                per[name].append(a.elapsed_time(b) * 1e3)
    for k in forms:
        res[k].append(round(sorted(per[k])[len(per[k]) // 2], 1))
print(json.dumps({"workload": what, "us_per_view": res, "mean_us": {k: round(sum(v) / len(v), 1) for k, v in res.items()}}))
