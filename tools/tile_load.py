"""Developer tool: how is K7's work spread over the tiles?  Per tile: list length n, deepest stop position tmax (what K7
walks), and the share of the launch's critical path: sum(tmax) / wave slots against the longest single tile.
    python tools/tile_load.py [N W H clustered(0/1)]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
H = int(sys.argv[3]) if len(sys.argv) > 3 else 720
cl = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
deg = 3
dev = torch.device("cuda:0")
P, intr = synthetic_gaussians(N, W, H, deg, 78, clustered=cl)
D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
out = []
for view in range(0, 8, 2):
    cam = make_camera(intr, view, 8)
    splats, radii, group_base, tile_start, sorted_gid, status = ops.project_bin_sort(
        cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
    rgb, depth, fT, _ = ops.rasterize_fwd(cam, splats, sorted_gid, tile_start)
    ts = tile_start[:cam.num_tiles + 1].long()
    n = (ts[1:] - ts[:-1])
    TW, TH = (W + 15) // 16, (H + 15) // 16
    sp = fT.stop_pos
    pad = torch.zeros(TH * 16, TW * 16, dtype=torch.int64, device=dev)
    pad[:H, :W] = sp.long().clamp(max=int(n.max()))
    tmax = pad.view(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(TH * TW, 256).max(1).values
    tmax = torch.minimum(tmax, n)
    slots = 256 * 4 * 4          # CUs x SIMDs x K7 waves per SIMD
    q = lambda t, p: int(torch.quantile(t.float(), p))
    out.append({"view": view, "pairs": int(n.sum()), "walked": int(tmax.sum()), "n_max": int(n.max()), "n_q99": q(n, 0.99),
                "tmax_max": int(tmax.max()), "tmax_q99": q(tmax, 0.99), "tmax_mean": round(float(tmax.float().mean()), 1),
                "tiles": int(n.numel()), "tiles_nonempty": int((n > 0).sum()),
                "balanced_walk_per_slot": round(float(tmax.sum()) / slots, 1),
                "critical_over_balanced": round(float(tmax.max()) / (float(tmax.sum()) / slots), 2)})
print(json.dumps({"N": N, "W": W, "H": H, "clustered": cl, "views": out}, indent=1))
