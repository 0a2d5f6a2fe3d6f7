"""Developer tool: how many blend iterations the 4x4-block form of K6 runs against the quadrant form (cfg3,
view 0, a sample of tiles; liveness ignored in both): per batch of 64 list entries the quadrant form evaluates
|Gaussians reaching the quadrant| per quadrant, the block form max(|Gaussians reaching block|) over the
quadrant's four blocks (rounded up to 4)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians
from tests.util import splat_fields
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device("cuda:0")
P, intr = synthetic_gaussians(N, W, H, deg, 1236)
cam = make_camera(intr, 0, 8)
D = {k: v.to(dev).contiguous() for k, v in P.items()}
sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
f = splat_fields(sp)
xy, conic, opac = f["xy"].numpy(), f["conic"].numpy(), f["opac"].numpy()
ts_h, sg_h = ts.cpu().numpy(), sg.cpu().numpy()
TW = (W + 15) // 16
rng = np.random.default_rng(0)
tiles = rng.choice(len(ts_h) - 1, 400, replace=False)
n_old = n_new = n_new_exact = n_pairs = 0
n_mean = 0.0
n128 = 0
for t in tiles:
    s, e = ts_h[t], ts_h[t + 1]
    ty, tx = divmod(t, TW)
    ys, xs = np.mgrid[0:16, 0:16]
    px, py = tx * 16 + xs + 0.5, ty * 16 + ys + 0.5
    for b0 in range(s, e, 64):
        g = sg_h[b0:min(b0 + 64, e)]
        dx = xy[g, 0][:, None, None] - px[None]; dy = xy[g, 1][:, None, None] - py[None]
        a, b, c = (conic[g, i][:, None, None] for i in range(3))
        al = opac[g][:, None, None] * np.exp(-(0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy))
        reach = al >= 1.0 / 255.0                                  # [n,16,16]
        blk = reach.reshape(len(g), 4, 4, 4, 4).any(axis=(2, 4))   # [n, by, bx]
        quad = blk.reshape(len(g), 2, 2, 2, 2).any(axis=(2, 4))    # [n, qy, qx]
        n_pairs += len(g)
        n_old += int(quad.sum())
        cnt = blk.sum(0)                                            # [by, bx]
        mx = cnt.reshape(2, 2, 2, 2).max(axis=(1, 3))               # per quadrant
        n_new_exact += int(mx.sum())
        n_new += int((((mx + 1) // 2) * 2).sum())
        n_mean += float(cnt.reshape(2, 2, 2, 2).mean(axis=(1, 3)).sum())
    for b0 in range(s, e, 128):
        g = sg_h[b0:min(b0 + 128, e)]
        dx = xy[g, 0][:, None, None] - px[None]; dy = xy[g, 1][:, None, None] - py[None]
        a, b, c = (conic[g, i][:, None, None] for i in range(3))
        al = opac[g][:, None, None] * np.exp(-(0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy))
        blk = (al >= 1.0 / 255.0).reshape(len(g), 4, 4, 4, 4).any(axis=(2, 4))
        mx = blk.sum(0).reshape(2, 2, 2, 2).max(axis=(1, 3))
        n128 += int((((mx + 1) // 2) * 2).sum())
print(f"pairs {n_pairs}: quadrant evaluations {n_old} ({n_old / n_pairs:.2f} per pair); block-form iterations {n_new_exact} "
      f"({n_new_exact / n_old:.3f} of the quadrant form), rounded up to 2: {n_new} ({n_new / n_old:.3f}); "
      f"if the four rows were perfectly balanced: {n_mean / n_old:.3f}; with batches of 128 (rounded to 2): {n128 / n_old:.3f}")
