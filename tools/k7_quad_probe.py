"""Developer tool: where does K7's four-wave form pay?  Per view: the frame's chain ratio (deepest walk / (sum of walks /
4096 slots)) and K7's time with one wave per tile and with four waves for every tile walking > min_walk entries.
    python tools/k7_quad_probe.py N W H clustered(0/1)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians

N, W, H, cl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
deg = 3
dev = torch.device("cuda:0")
P, intr = synthetic_gaussians(N, W, H, deg, 78, clustered=cl)
D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
g = torch.Generator().manual_seed(1)
v_rgb = torch.randn(H, W, 3, generator=g).to(dev)
v_d = torch.randn(H, W, generator=g).to(dev)
before = ops.set_k7_quad()
rows = []
for view in range(8):
    cam = make_camera(intr, view, 8)
    sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
    rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    n = (ts[1:cam.num_tiles + 1] - ts[:cam.num_tiles]).long()
    TW, TH = (W + 15) // 16, (H + 15) // 16
    pad = torch.zeros(TH * 16, TW * 16, dtype=torch.int64, device=dev)
    pad[:H, :W] = fT.stop_pos.long().clamp(max=int(n.max()))
    tmax = torch.minimum(pad.view(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(TH * TW, 256).max(1).values, n)
    ratio = float(tmax.max()) / max(float(tmax.sum()) / 4096, 1e-9)
    t = {}
    for name, f in (("one", 0), ("quad", 1)):
        ops.set_k7_quad(f, 48)
        for _ in range(3):
            ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, v_rgb=v_rgb, v_depth=v_d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, v_rgb=v_rgb, v_depth=v_d)
        e1.record(); torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) / 10 * 1e3
    rows.append(dict(view=view, walk_max=int(tmax.max()), walk_sum=int(tmax.sum()), ratio=round(ratio, 2),
                     k7_one_us=round(t["one"], 1), k7_quad_us=round(t["quad"], 1), quad_over_one=round(t["quad"] / t["one"], 3)))
    print(json.dumps(rows[-1]), flush=True)
ops.set_k7_quad(*before)
