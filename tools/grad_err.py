"""Developer tool: error distribution of the HIP raster backward's five parameter gradients against the fp64 C
oracle, split by the decision margin of the pixels a Gaussian reaches (the classification
tests/test_gpu_fullsize_oracle.py asserts on).   python tools/grad_err.py cfg3"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.ref_c import RefC
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians
from tests.test_gpu_fullsize_oracle import order_ambiguous_tiles
CONFIGS = {"cfg2": (100_000, 800, 800, 3, 1235, 1), "cfg3": (1_000_000, 1920, 1080, 3, 1236, 0),
           "cfg5": (5_000_000, 3840, 2160, 3, 1238, 3)}
N, W, H, deg, seed, view = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
dev = torch.device("cuda:0")
P, intr = synthetic_gaussians(N, W, H, deg, seed)
cam = make_camera(intr, view, 8, bg=(0.1, 0.2, 0.3))
D = {k: v.to(dev).contiguous() for k, v in P.items()}
sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, want_radii=True)
n_hip = st.tolist()[0]
rgb, depth, fT, fidx = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=True)
g = torch.Generator().manual_seed(seed)
v_rgb = torch.randn(H, W, 3, generator=g); v_d = torch.randn(H, W, generator=g); v_a = torch.randn(H, W, generator=g)
partials, _ = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, v_rgb.to(dev), v_d.to(dev), v_a.to(dev))
grads = ops.project_bwd(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, sp, gb, partials)
R = RefC("f64")
f32 = lambda v: float(np.float32(v))
n64 = lambda t: t.double().numpy()
cb = R.cam_block(np.asarray(cam.viewmat, np.float32).astype(np.float64).reshape(4, 4), f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy), bg=tuple(f32(c) for c in cam.bg))
Pn = {k: n64(v) for k, v in P.items()}
pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
g2, ts2 = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
margin = R.blend_margin(pc["xy"], pc["conic"], pc["opac"], g2, ts2, cb, W, H)
bb = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H, bf["final_T"], bf["final_idx"], n64(v_rgb), n64(v_d), n64(v_a))
pb = R.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H, pc["radius"], bb["v_xy"], bb["v_conic"], bb["v_opac"], bb["v_rgb"], bb["v_depth"])
TW = (W + 15) // 16
yy, xx = np.mgrid[0:H, 0:W]
tile_px = (yy // 16) * TW + xx // 16
bad = order_ambiguous_tiles(sg.cpu().numpy()[:n_hip], ts.cpu().numpy(), g2, ts2, N)
pm = margin.copy()
pm[np.isin(tile_px, bad)] = 0.0
print(f"{sys.argv[1:]}: order-ambiguous tiles {len(bad)}; pixel margins: " + " ".join(f">{t:g}: {np.mean(pm > t):.5f}" for t in (1e-3, 1e-4, 3e-5, 1e-5, 3e-6)))
gmin, npix = R.gaussian_min_margin(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], cb, W, H, pm, near=0.5)
touch = npix > 0
print(f"Gaussians reaching some pixel: {touch.mean():.4f}; pixels reached, mean {npix[touch].mean():.1f}")
names = ("means", "log_scales", "quats", "opac_logit", "sh")
keys = ("v_means", "v_log_scales", "v_quats", "v_opac_logit", "v_sh")
for thr in (1e-3, 1e-4, 3e-5, 1e-5, 3e-6, 1e-6):
    clear = touch & (gmin > thr)
    print(f"--- Gaussian margin > {thr:g}: {clear.sum() / max(touch.sum(), 1):.4f} of the reaching Gaussians")
    for name, got, key in zip(names, grads[:5], keys):
        got = got.cpu().double().numpy().reshape(N, -1)
        ref = pb[key].reshape(N, -1)
        # per-Gaussian error relative to the Gaussian's own gradient norm, with a floor relative to the typical norm
        nrm = np.abs(ref).max(1)
        typ = np.median(nrm[touch & (nrm > 0)])
        err = np.abs(got - ref).max(1)
        out = []
        for fl in (1e-1, 1e-2, 1e-3):
            e = err / (nrm + fl * typ)
            ec, eu = e[clear], e[touch & ~clear]
            out.append(f"floor {fl:g}*typ: clear max {ec.max():.2e} q99.99 {np.quantile(ec, 0.9999):.2e} q99 {np.quantile(ec, 0.99):.2e} n>1e-4 {int((ec > 1e-4).sum())}"
                       + (f" | unclear q50 {np.median(eu):.1e} q99 {np.quantile(eu, 0.99):.1e} max {eu.max():.1e} frac>1e-4 {np.mean(eu > 1e-4):.3f}" if len(eu) else ""))
        print(f"  {name:11s} typ {typ:.2e}  " + "\n              ".join(out))
# Gaussians that reach nothing must have zero gradient in both
for name, got, key in zip(names, grads[:5], keys):
    got = got.cpu().double().numpy().reshape(N, -1); ref = pb[key].reshape(N, -1)
    print(f"non-reaching: {name} max |got| {np.abs(got[~touch]).max():.2e} max |ref| {np.abs(ref[~touch]).max():.2e}")
