"""Developer tool: error distribution of the HIP forward against the fp64 C oracle at one BASELINE config
(the quantities tests/test_gpu_fullsize_oracle.py asserts on).   python tools/fullsize_err.py cfg5"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.ref_c import RefC
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians
from tests.util import relerr
CONFIGS = {"cfg2": (100_000, 800, 800, 3, 1235, 1), "cfg3": (1_000_000, 1920, 1080, 3, 1236, 0),
           "cfg5": (5_000_000, 3840, 2160, 3, 1238, 3)}
N, W, H, deg, seed, view = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg5"]
dev = torch.device("cuda:0")
P, intr = synthetic_gaussians(N, W, H, deg, seed)
cam = make_camera(intr, view, 8, bg=(0.1, 0.2, 0.3))
D = {k: v.to(dev).contiguous() for k, v in P.items()}
sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, want_radii=True)
rgb, depth, fT, fidx = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=True)
R = RefC("f64")
f32 = lambda v: float(np.float32(v))
cb = R.cam_block(np.asarray(cam.viewmat, np.float32).astype(np.float64).reshape(4, 4), f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy), bg=tuple(f32(c) for c in cam.bg))
Pn = {k: v.double().numpy() for k, v in P.items()}
pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
g2, ts2 = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
margin = R.blend_margin(pc["xy"], pc["conic"], pc["opac"], g2, ts2, cb, W, H)
er = relerr(rgb.cpu().numpy(), bf["rgb"], floor=1e-2).max(-1)
ed = relerr(depth.cpu().numpy(), bf["depth_acc"], floor=1e-2)
eT = np.abs(fT.cpu().numpy() - bf["final_T"])
for thr in (1e-3, 3e-3, 1e-2):
    clear = margin > thr
    print(f"margin > {thr}: {clear.mean():.4f} of the pixels")
    for name, e in (("rgb", er), ("depth", ed), ("final_T", eT)):
        v = e[clear]
        qs = [np.quantile(v, q) for q in (0.5, 0.99, 0.999, 0.9999, 0.99999)]
        print(f"  {name:8s} frac<1e-4 {np.mean(v < 1e-4):.6f}  n_viol {int((v >= 1e-4).sum())}  q50/99/99.9/99.99/99.999 " +
              " ".join(f"{q:.2e}" for q in qs) + f"  max {v.max():.2e}")
# record xy against the oracle's
from tests.util import splat_fields
f = splat_fields(sp, radii)
vis = (f["radius"].numpy() > 0) & (pc["radius"] > 0)
dxy = np.abs(f["xy"].numpy() - pc["xy"])[vis]
print("record xy vs oracle: median %.2e  q99.9 %.2e  max %.2e px" % (np.median(dxy), np.quantile(dxy, 0.999), dxy.max()))
dc = np.abs(f["conic"].numpy() - pc["conic"])[vis] / (np.abs(pc["conic"][vis]).max(-1, keepdims=True) + 1e-30)
print("record conic rel err: median %.2e  q99.9 %.2e  q99.999 %.2e  max %.2e" % (np.median(dc), np.quantile(dc, 0.999), np.quantile(dc, 0.99999), dc.max()))
# ---- are the outliers depth-order ambiguities?  (fp32 depth keys order two nearly equal depths differently
#      from the fp64 oracle: the two colours swap, T and the depth sum stay)
sg_h = sg.cpu().numpy().astype(np.int64)[: int(st.tolist()[0])]
ts_h = ts.cpu().numpy().astype(np.int64)
T_ = len(ts_h) - 1
tile_h = np.repeat(np.arange(T_), np.diff(ts_h))
tile_o = np.repeat(np.arange(T_), np.diff(ts2.astype(np.int64)))
key_o = tile_o * N + g2.astype(np.int64)
order_o = np.argsort(key_o, kind="stable")
pos_in_o = np.searchsorted(key_o[order_o], tile_h * N + sg_h)
found = key_o[order_o][np.minimum(pos_in_o, len(key_o) - 1)] == tile_h * N + sg_h
rank_o = order_o[np.minimum(pos_in_o, len(key_o) - 1)]      # position of the HIP pair in the oracle's global list
same_tile = tile_h[1:] == tile_h[:-1]
inv = same_tile & found[1:] & found[:-1] & (rank_o[1:] < rank_o[:-1])
bad_tiles = np.unique(tile_h[1:][inv])
print(f"tiles whose HIP list orders some pair differently from the fp64 oracle: {len(bad_tiles)} of {T_}")
TW = (W + 15) // 16
yy, xx = np.mgrid[0:H, 0:W]
tile_px = (yy // 16) * TW + xx // 16
in_bad = np.isin(tile_px, bad_tiles)
clear = margin > 1e-3
viol = (er >= 1e-4) & clear
print(f"rgb violators: {int(viol.sum())}, of which inside such tiles: {int((viol & in_bad).sum())}; "
      f"max rgb err outside: {er[clear & ~in_bad].max():.2e}  depth {ed[clear & ~in_bad].max():.2e}  T {eT[clear & ~in_bad].max():.2e}")
