"""Developer tool: where the geometry-gradient error of the HIP backward comes from.  Compares the per-Gaussian
screen-space gradients K7 produces (v_xy, v_conic, v_opacity, v_rgb, v_depth = tgs_reduce_partials of K7's
partial records) with the fp64 C oracle's blend backward, next to the fp32 build of the same scalar oracle
(plain back-to-front fp32 accumulation of v_sigma * dx * dx etc. -- what a CUDA rasterizer computes).
   python tools/grad_err_k7.py cfg3"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle.ref_c import RefC
from touch_gs_amd import ops
from touch_gs_amd.scene import make_camera, synthetic_gaussians
from tests.test_gpu_fullsize_oracle import order_ambiguous_tiles
CONFIGS = {"cfg2": (100_000, 800, 800, 3, 1235, 1), "cfg3": (1_000_000, 1920, 1080, 3, 1236, 0)}
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
vs = ops.reduce_partials(cam, sp, gb, partials).cpu().double().numpy()
hip = dict(v_xy=vs[:, 0:2], v_depth=vs[:, 2:3], v_opac=vs[:, 3:4], v_conic=vs[:, 4:7], v_rgb=vs[:, 7:10])
f32 = lambda v: float(np.float32(v))
n64 = lambda t: t.double().numpy()
Pn = {k: n64(v) for k, v in P.items()}
out = {}
for prec in ("f64", "f32"):
    R = RefC(prec)
    cb = R.cam_block(np.asarray(cam.viewmat, np.float32).astype(np.float64).reshape(4, 4), f32(cam.fx), f32(cam.fy), f32(cam.cx), f32(cam.cy), bg=tuple(f32(c) for c in cam.bg))
    pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
    if prec == "f64":
        pc64, cb64, R64 = pc, cb, R
        g2, ts2 = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
    else:   # the fp32 build blends the fp64 projection's records (rounded): isolates the blend backward
        pc = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in pc64.items()}
    bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
    bb = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H, bf["final_T"], bf["final_idx"], n64(v_rgb), n64(v_d), n64(v_a))
    out[prec] = {k: np.asarray(v, np.float64).reshape(N, -1) for k, v in bb.items()}
margin = R64.blend_margin(pc64["xy"], pc64["conic"], pc64["opac"], g2, ts2, cb64, W, H)
TW = (W + 15) // 16
yy, xx = np.mgrid[0:H, 0:W]
bad = order_ambiguous_tiles(sg.cpu().numpy()[:n_hip], ts.cpu().numpy(), g2, ts2, N)
pm = margin.copy(); pm[np.isin((yy // 16) * TW + xx // 16, bad)] = 0.0
gmin, npix = R64.gaussian_min_margin(pc64["xy"], pc64["conic"], pc64["opac"], pc64["rect"], pc64["tiles_hit"], cb64, W, H, pm, near=0.5)
clear = (npix > 0) & (gmin > 1e-4)
print(f"{sys.argv[1:]}: clear Gaussians (margin > 1e-4): {clear.sum()} of {(npix > 0).sum()} reaching")
for key in ("v_rgb", "v_depth", "v_opac", "v_xy", "v_conic"):
    ref = out["f64"][key]
    nrm = np.abs(ref).max(1); typ = np.median(nrm[clear & (nrm > 0)])
    line = f"{key:8s} typ {typ:.2e} "
    for name, got in (("HIP", hip[key]), ("C-f32", out["f32"][key])):
        e = (np.abs(got - ref).max(1) / (nrm + 0.1 * typ))[clear]
        line += f"| {name}: q50 {np.median(e):.1e} q99 {np.quantile(e, 0.99):.1e} q99.99 {np.quantile(e, 0.9999):.1e} max {e.max():.1e} "
    print(line)
# ---- error against the un-cancelled magnitude of each gradient (oracle: blend_bwd(mass=True)) ----
bf64 = R64.blend_fwd(pc64["xy"], pc64["conic"], pc64["opac"], pc64["rgb"], pc64["depth"], g2, ts2, cb64, W, H)
mass = R64.blend_bwd(pc64["xy"], pc64["conic"], pc64["opac"], pc64["rgb"], pc64["depth"], g2, ts2, cb64, W, H, bf64["final_T"], bf64["final_idx"], n64(v_rgb), n64(v_d), n64(v_a), mass=True)
print("--- |HIP - oracle| / un-cancelled magnitude, clear Gaussians (margin > 1e-4 / > 1e-5 / all reaching):")
for key in ("v_rgb", "v_depth", "v_opac", "v_xy", "v_conic"):
    ref, m = out["f64"][key], np.asarray(mass[key], np.float64).reshape(N, -1)
    assert (m >= np.abs(ref) * (1 - 1e-12)).all()
    e = (np.abs(hip[key] - ref) / (m + 1e-300)).max(1)
    line = f"{key:8s}"
    for sel in (clear, (npix > 0) & (gmin > 1e-5), npix > 0):
        v = e[sel]
        line += f" | n {sel.sum()} q50 {np.median(v):.1e} q99.99 {np.quantile(v, 0.9999):.1e} max {v.max():.1e} n>1e-4 {int((v > 1e-4).sum())}"
    print(line)
# ---- K8 in isolation: the oracle's projection backward on HIP's own K7 output against HIP's K8 ----
grads = ops.project_bwd(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, sp, gb, partials)
pb = R64.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb64, W, H, pc64["radius"],
                     hip["v_xy"], hip["v_conic"], hip["v_opac"][:, 0], hip["v_rgb"], hip["v_depth"][:, 0])
print("--- K8 alone (oracle project_bwd fed HIP's K7 output), error / (per-Gaussian max-norm + 1e-3 typ), all Gaussians:")
for name, got, key in zip(("means", "log_scales", "quats", "opac_logit", "sh"), grads[:5], ("v_means", "v_log_scales", "v_quats", "v_opac_logit", "v_sh")):
    got = got.cpu().double().numpy().reshape(N, -1); ref = pb[key].reshape(N, -1)
    nrm = np.abs(ref).max(1); typ = np.median(nrm[nrm > 0])
    e = np.abs(got - ref).max(1) / (nrm + 1e-3 * typ)
    vis = (pc64["radius"] > 0)
    print(f"{name:11s} q50 {np.median(e[vis]):.1e} q99.99 {np.quantile(e[vis], 0.9999):.1e} max {e[vis].max():.1e} n>1e-4 {int((e[vis] > 1e-4).sum())}  (invisible: max |got| {np.abs(got[~vis]).max():.1e})")
# ---- worst v_conic cases among the clear Gaussians ----
ref, m = out["f64"]["v_conic"], np.asarray(mass["v_conic"], np.float64).reshape(N, -1)
e = (np.abs(hip["v_conic"] - ref) / (m + 1e-300)).max(1) * clear
for i in np.argsort(-e)[:8]:
    print(f"g {i}: err/mass {e[i]:.2e} radius {pc64['radius'][i]} rect {pc64['rect'][i]} xy {pc64['xy'][i]} opac {pc64['opac'][i]:.4f} conic {pc64['conic'][i]} npix {npix[i]}\n"
          f"     ref {ref[i]} hip {hip['v_conic'][i]} mass {m[i]}  v_opac ref {out['f64']['v_opac'][i]} hip {hip['v_opac'][i]}")
