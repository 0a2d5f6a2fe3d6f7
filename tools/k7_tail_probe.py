"""Developer tool: is K7's four-wave launch bound by its longest tile?  Per training view of a 720p checkpoint: deepest walk, sum of walks,
tiles walking > 16 entries, and the times of the two K7 launches (events)."""
import json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from touch_gs_amd import ops
import ckpt_loop
what = sys.argv[1] if len(sys.argv) > 1 else "bunny"
if len(sys.argv) > 2:
    ops.set_k7_scan(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 512)
dev = torch.device("cuda:0")
m, views = ckpt_loop.load(os.path.join(ROOT, "build_ab/ckpt/model_%s_1.pt" % ("bunny_real" if what == "bunny" else "block")), dev, n_views=8)
p, deg = m.params, 3
rows = []
for v in views:
    cam = v.cam; H, W = cam.H, cam.W; T = cam.num_tiles; TW, TH = cam.tiles
    ts_ = []
    for rep in range(4):
        sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
        rgb, dacc, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
        ss, vimg = ops.ssim_fwd_bwd(rgb, v.rgb, weight=-0.2 / (3 * H * W), reduce=False)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=m.loss_spec(v), want_tile_loss=True); b.record()
        torch.cuda.synchronize()
        if rep: ts_.append(a.elapsed_time(b) * 1e3)
    n = (ts[1:T + 1] - ts[:T]).long()
    pad = torch.zeros(TH * 16, TW * 16, dtype=torch.int64, device=dev)
    pad[:H, :W] = fT.stop_pos.long().clamp(max=int(n.max()))
    walk = torch.minimum(pad.view(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(T, 256).max(1).values, n)
    q = torch.sort(walk, descending=True).values
    rows.append(dict(k7_us=round(sorted(ts_)[1], 1), walk_max=int(q[0]), walk_10th=int(q[9]), walk_100th=int(q[99]), walk_sum=int(walk.sum()),
                     tiles_over_16=int((walk > 16).sum()), longest_list=int(n.max()), pairs=int(n.sum())))
    print(json.dumps(rows[-1]), flush=True)
