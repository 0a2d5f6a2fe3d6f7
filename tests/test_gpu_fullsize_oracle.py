"""Full-size oracle parity on the BASELINE configs: the HIP path against the
fp64 build of the scalar C oracle (oracle/ref_raster.c) on

* configs[1]: 100 k Gaussians, 800x800   (seed 1235),
* configs[2]: 1 M Gaussians, 1920x1080   (seed 1236),
* configs[4], the part one GPU runs: 5 M Gaussians, SH degree 3, 3840x2160 (seed 1238),

forward (RGB / depth / final T / last contributor) on every pixel whose threshold decisions are not
ambiguous under fp32 rounding, all five parameter gradients, the FULL train step of configs[1] / [2]
(fused L1 + tactile depth/uncertainty loss + SSIM with the reference's three flag sets: loss value and
parameter gradients against O.train_loss differentiated through the C oracle), SSIM value + gradient
image at the three image sizes (covers every segment length k_ssim_* selects), and the
output-preservation claim of the tight tile rectangle: every (tile, Gaussian) pair that the product drops
from the NORMATIVE App. B.4 rectangle has alpha < 1/255 at every pixel centre of its tile.

The oracle lists are built from the normative B.4 rect, the HIP lists from the tight rect -- the
comparison therefore also proves that the dropped pairs do not change the image or the gradients.
The oracle needs the GPU box's host cores (~128): a few seconds per config.  Parity remains
UNPINNED against the reference rasterizer (source absent, see oracle/__init__.py).
"""
import numpy as np
import pytest
import torch

from oracle.ref_c import RefC
from tests.util import rect_from, relerr, splat_fields

pytestmark = pytest.mark.gpu

CONFIGS = {"cfg2_100k_800x800": (100_000, 800, 800, 3, 1235, 1),
           "cfg3_1M_1080p": (1_000_000, 1920, 1080, 3, 1236, 0),
           # the per-GPU part of configs[4] (5 M Gaussians, SH degree 3, 4K render; seed 1238, view 3 of 8)
           "cfg5_5M_4K": (5_000_000, 3840, 2160, 3, 1238, 3)}
TOL = 1e-4
# fraction of the pixels with every threshold decision further than 1e-3 (relative) from its threshold and an
# fp32 depth order equal to the fp64 one -- where RGB / depth / T are asserted at 1e-4 (r4 measurements)
CLEAR_MEASURED = {"cfg2_100k_800x800": 0.987, "cfg3_1M_1080p": 0.960, "cfg5_5M_4K": 0.91}
# share of the REACHING Gaussians that are decision-clear -- the population the gradient bar (max <= 1e-4) is asserted on
# -- as measured (profiles/r4_gradient_parity_fullsize.txt, r6_gradient_parity_fullsize.txt); asserted within +- 0.02 so
# that a drift of the classification fails instead of silently shrinking the asserted population (VERDICT r5 next #4b).
# "train": the fused train step's classification (pixels whose L1 sign is within the forward's tolerance count as ambiguous)
CLEAR_GAUSSIANS_MEASURED = {"cfg2_100k_800x800": 0.7755, "cfg3_1M_1080p": 0.5170, "cfg5_5M_4K": 0.61}
CLEAR_GAUSSIANS_MEASURED_TRAIN = {"cfg2_100k_800x800": 0.5425, "cfg3_1M_1080p": 0.3802, "cfg5_5M_4K": None}
REPORT_FILE = "gpurun_out/r6_gradient_parity_fullsize.txt"     # the grad_report lines, for the next reader (VERDICT r5 next #4d)


def _write_report(lines):
    import os
    os.makedirs(os.path.dirname(REPORT_FILE), exist_ok=True)
    with open(REPORT_FILE, "a") as f:
        f.write("\n".join(lines) + "\n")


# Not BASELINE configs: scenes that put one code path under the same oracle machinery (their own tests below).
#   huge_720p: the reference's image size with a dozen SCREEN-FILLING Gaussians (a table / background Gaussian of a
#   converged object-centric model covers 1000 - 3600 tiles): K8's workgroup-shared segmented sum of long runs, the
#   direct counting path and k_fill_bins on groups of tens of thousands of pairs (round 6).
#   clamp_640: every third Gaussian with opacity > 0.999 and e^2 larger axes: tens of thousands of (pixel, Gaussian) pairs ON
#   the alpha = 0.999 clamp (App. B.6 / B.7), SH degree 2, at a size where the strict bar of the full-size test can be asserted in both
#   forms of K7 (the small-size test of test_gpu_parity.py is statistical).
#   hard_640: strata the random BASELINE scenes do not hold -- needles (16 : 1), sub-pixel Gaussians (the 0.3 px blur is
#   most of their footprint), Gaussians 3 - 30 cm from the camera, never-visible faint ones, SH bands large enough to drive
#   colours through the zero clamp, and a stratum 20x farther away at 20x the size (same screen footprint, large depths).
EXTRA_CONFIGS = {"huge_720p": (30_000, 1280, 720, 3, 77, 1), "clamp_640": (20_000, 640, 400, 2, 91, 1),
                 "hard_640": (24_000, 640, 400, 3, 93, 0),
                 # every other scene of the suite has fx = fy and the principal point in the image centre; a RealSense
                 # (the reference's camera, utils/read_realsense_depth.py) has neither.  637 x 395: ragged right / bottom tiles
                 "skewed_640": (20_000, 637, 395, 3, 95, 1)}
HUGE_EVERY = 2500     # rows 0, 2500, 5000, ... of huge_720p are blown up


def _mutate_huge(P):
    idx = torch.arange(0, P["means"].shape[0], HUGE_EVERY)
    # x 150, sigma ~ 1000 px on every axis (not isotropic: the quaternion gradient of three equal scales vanishes
    # analytically and K8's fp32 chain then rounds at the size of its intermediates -- max_over(), exception (b))
    P["log_scales"][idx] = P["log_scales"][idx].mean(1, keepdim=True) + 5.0 + torch.tensor([0.0, 0.3, -0.3]).to(P["log_scales"])
    P["opac_logit"][idx] = -2.0            # faint (0.12), so that the scene behind them still contributes
    P["means"][idx, 2] = P["means"][idx, 2].clamp(min=3.0)


def _mutate_clamp(P):
    P["opac_logit"][::3] = 12.0
    P["log_scales"][::3] += 2.0 + torch.tensor([0.0, 0.2, -0.2]).to(P["log_scales"])    # (not isotropic: max_over(), exception (b))


def _mutate_hard(P):
    N = P["means"].shape[0]
    i = torch.arange(N)
    g = torch.Generator().manual_seed(5)
    needle, tiny, near, faint, loud, far = (i % 8 == k for k in range(6))
    # 16 : 1.  (Not more: the conic is the inverse of a 2x2 covariance whose condition number is the squared aspect ratio;
    # any fp32 evaluation, the reference's included, carries eps x aspect^2 into it -- measured at 90 : 1: conics off by
    # 4.5e-4, pixels by 2e-4 -- so the 1e-4 bar ends near 40 : 1.)
    P["log_scales"][needle] = P["log_scales"][needle].mean(1, keepdim=True) + torch.tensor([1.6, -1.2, -1.2]).to(P["log_scales"])
    P["log_scales"][tiny] -= 3.0
    z_new = 0.03 + 0.3 * torch.rand(int(near.sum()), generator=g)           # view 0: camera z = world z
    k = (z_new / P["means"][near, 2]).to(P["means"])
    P["means"][near] = P["means"][near] * k[:, None]                          # same pixel, closer
    P["log_scales"][near] += torch.log(k)[:, None]                            # same footprint
    P["opac_logit"][faint] = -7.0                                             # alpha < 1 / 255 everywhere
    P["sh"][loud, 1:, :] *= 20.0
    P["means"][far] *= 20.0
    P["log_scales"][far] += float(np.log(20.0))


EXTRA_MUTATE = {"huge_720p": _mutate_huge, "clamp_640": _mutate_clamp, "hard_640": _mutate_hard}
# fx 15 % longer than fy, principal point 37 px right of and 21.5 px above the centre (the image no longer sits in the
# middle of the frustum the 1.3 x tan(fov / 2) clamp of B.2 is built around)
EXTRA_INTRINSICS = {"skewed_640": lambda i: dict(fx=1.15 * i["fx"], fy=i["fy"], cx=i["cx"] + 37.0, cy=i["cy"] - 21.5)}


def order_ambiguous_tiles(sg_hip, ts_hip, g_ref, ts_ref, N):
    """Tiles in which the product's list orders some pair of Gaussians differently from the fp64 oracle's.
    The lists are sorted by the fp32 depth bits (as the CUDA rasterizers behind the reference sort fp32 depth
    keys); two depths closer than fp32 resolves can compare the other way round, or tie and fall back to the
    id, in fp64.  Swapping two neighbours of nearly the same depth swaps their colours (final T and the depth
    sum are unaffected): a property of fp32 depth keys, not of the compositing -- such tiles (0.5 % at 4K / 5 M
    Gaussians, ~700 per tile) are excluded from the 1e-4 comparison and counted."""
    sg_hip, g_ref = sg_hip.astype(np.int64), g_ref.astype(np.int64)
    ts_hip, ts_ref = ts_hip.astype(np.int64), ts_ref.astype(np.int64)
    T = len(ts_hip) - 1
    tile_h = np.repeat(np.arange(T), np.diff(ts_hip))
    tile_o = np.repeat(np.arange(T), np.diff(ts_ref))
    key_o = tile_o * N + g_ref
    order_o = np.argsort(key_o, kind="stable")
    ks = key_o[order_o]
    key_h = tile_h * N + sg_hip
    pos = np.minimum(np.searchsorted(ks, key_h), len(ks) - 1)
    found = ks[pos] == key_h                       # (the product's lists are a subset of the normative ones)
    rank_o = order_o[pos][found]                   # position of every product pair in the oracle's global list
    th = tile_h[found]
    inv = (th[1:] == th[:-1]) & (rank_o[1:] < rank_o[:-1])
    # ... and tiles whose product list holds a Gaussian the oracle's list does not: the NORMATIVE B.4 rect is an
    # integer decision (ceil of the 3-sigma radius, truncation of (u -+ r) / 16) that fp32 takes differently from
    # fp64 for ~0.5 % of the Gaussians, and for opacities near 1 the region alpha >= 1/255 (3.33 sigma) reaches
    # beyond the 3-sigma rect -- the extra list member then contributes on one side only
    return np.unique(np.concatenate([th[1:][inv], tile_h[~found]]))


def missing_pairs(sg_hip, ts_hip, g_ref, ts_ref, N):
    """(gid, tile) of the pairs of the oracle's lists (normative B.4 rect) that the product's lists do not hold:
    dropped by the tight rect (or by an fp32 rect that is a tile smaller)."""
    sg_hip, g_ref = sg_hip.astype(np.int64), g_ref.astype(np.int64)
    T = len(ts_hip) - 1
    key_h = np.repeat(np.arange(T), np.diff(ts_hip.astype(np.int64))) * N + sg_hip
    tile_o = np.repeat(np.arange(T), np.diff(ts_ref.astype(np.int64)))
    key_o = tile_o * N + g_ref
    miss = ~np.isin(key_o, key_h)
    return g_ref[miss], tile_o[miss]


@pytest.fixture(scope="module", params=list(CONFIGS))
def both(request, dev):
    return build_case(request.param, dev)


def build_case(name, dev):
    """HIP path and fp64 C oracle on one BASELINE config (also used by tools/grad_offenders.py)."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg, seed, view = (CONFIGS.get(name) or EXTRA_CONFIGS[name])[:6]
    P, intr = synthetic_gaussians(N, W, H, deg, seed)
    if name in EXTRA_MUTATE:
        EXTRA_MUTATE[name](P)
    if name in EXTRA_INTRINSICS:
        intr.update(EXTRA_INTRINSICS[name](intr))
    cam = make_camera(intr, view, 8, bg=(0.1, 0.2, 0.3))
    D = {k: v.to(dev).contiguous() for k, v in P.items()}
    # ---- HIP ----
    sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                     D["sh"], deg, want_radii=True)
    n_hip = n_index = st.tolist()[0]
    rgb, depth, fT, fidx = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=True)
    g = torch.Generator().manual_seed(seed)
    v_rgb = torch.randn(H, W, 3, generator=g)
    v_d = torch.randn(H, W, generator=g)
    v_a = torch.randn(H, W, generator=g)
    partials, _ = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, v_rgb.to(dev), v_d.to(dev), v_a.to(dev))
    grads = ops.project_bwd(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, sp, gb, partials)
    # last contributor as a Gaussian id (list positions differ between tight and normative lists)
    TW = (W + 15) // 16
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    tile = (yy // 16) * TW + xx // 16
    pos = ts.long()[tile] + fidx.long().clamp_min(0)
    last_gid_hip = torch.where(fidx >= 0, sg.long()[pos.clamp_max(max(n_hip - 1, 0))], torch.full_like(pos, -1)).cpu().numpy()
    # ---- oracle (fp64, normative B.4 lists) ----
    R = RefC("f64")
    n64 = lambda t: t.double().numpy()
    # identical inputs: the C ABI takes the camera as fp32 (TgsCamera), so the oracle gets exactly those values
    f32 = lambda v: float(np.float32(v))
    cb = R.cam_block(np.asarray(cam.viewmat, np.float32).astype(np.float64).reshape(4, 4), f32(cam.fx), f32(cam.fy),
                     f32(cam.cx), f32(cam.cy), bg=tuple(f32(c) for c in cam.bg))
    Pn = {k: n64(v) for k, v in P.items()}
    pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
    g2, ts2 = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
    bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
    margin = R.blend_margin(pc["xy"], pc["conic"], pc["opac"], g2, ts2, cb, W, H)
    bb = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H,
                     bf["final_T"], bf["final_idx"], n64(v_rgb), n64(v_d), n64(v_a))
    pb = R.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H,
                       pc["radius"], bb["v_xy"], bb["v_conic"], bb["v_opac"], bb["v_rgb"], bb["v_depth"])
    # un-cancelled magnitudes of the same gradients (the scale a floating-point sum is accurate against)
    m7 = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H,
                     bf["final_T"], bf["final_idx"], n64(v_rgb), n64(v_d), n64(v_a), mass=True)
    v_splats = ops.reduce_partials(cam, sp, gb, partials).cpu().double().numpy()
    tile_np = tile.cpu().numpy()
    pos2 = ts2[tile_np] + np.maximum(bf["final_idx"], 0)
    last_gid_ref = np.where(bf["final_idx"] >= 0, g2[np.minimum(pos2, max(len(g2) - 1, 0))], -1)
    order_ok = ~np.isin(tile_np, order_ambiguous_tiles(sg.cpu().numpy()[:n_hip], ts.cpu().numpy(), g2, ts2, N))
    # the converse: pairs of the oracle's list that the product's rect drops although alpha reaches 1/255 (less the
    # decision margin) at a pixel centre of the tile (the tight rect's fp32 extent falling short, or an fp32
    # normative rect one tile smaller): every tile of such a Gaussian's normative rect counts as ambiguous (a
    # handful of Gaussians per million; test_tight_rect_drops_only_invisible_pairs bounds their number)
    f = splat_fields(sp, radii)
    tight = f["rect"].numpy().astype(np.int32)
    tight[(f["hits"] == 0).numpy()] = 0
    mx, dropped = R.dropped_pairs_max_alpha(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], tight, cb, W, H)
    # ... pair by pair for everything the product's lists lack
    mg, mt = missing_pairs(sg.cpu().numpy()[:n_hip], ts.cpu().numpy(), g2, ts2, N)
    miss_alpha = R.pairs_max_alpha(mg, mt, pc["xy"], pc["conic"], pc["opac"], cb, W, H)
    for t in np.unique(mt[miss_alpha >= (1.0 - 1e-3) / 255.0]):
        order_ok[16 * (t // TW):16 * (t // TW) + 16, 16 * (t % TW):16 * (t % TW) + 16] = False
    return dict(name=name, N=N, W=W, H=H, deg=deg, seed=seed, view=view, cam=cam, D=D, Pn=Pn, g2=g2, ts2=ts2,
                order_ok=order_ok, dropped_max_alpha=mx, dropped=dropped, tight=tight, n_index=n_index,
                miss_gid=mg, miss_alpha=miss_alpha, bb=bb, m7=m7, v_splats=v_splats, cache={},
                R=R, cb=cb, sp=sp, radii=radii, n_hip=n_hip, n_ref=len(g2),
                rgb=rgb.cpu().numpy(), depth=depth.cpu().numpy(), fT=fT.cpu().numpy(), last_hip=last_gid_hip,
                grads=[t.cpu().double().numpy() for t in grads[:5]], pc=pc, bf=bf, margin=margin, pb=pb,
                last_ref=last_gid_ref)


def test_forward_matches_oracle_fullsize(both):
    b = both
    # pixels whose alpha_min / T_stop decisions are unambiguous under fp32 rounding, in tiles whose fp32 depth
    # order agrees with the fp64 order (order_ambiguous_tiles)
    assert b["order_ok"].mean() > 0.99, b["order_ok"].mean()
    clear = (b["margin"] > 1e-3) & b["order_ok"]
    # the share of the image the 1e-4 bound is asserted on, per config (measured: CLEAR_MEASURED; it cannot drift
    # by more than a percent without failing here)
    print(f"{b['name']}: decision-clear pixels (margin > 1e-3, depth order agrees) {clear.mean():.4f}")
    assert clear.mean() > CLEAR_MEASURED[b["name"]] - 0.01, (b["name"], clear.mean())
    er = relerr(b["rgb"], b["bf"]["rgb"], floor=1e-2)
    ed = relerr(b["depth"], b["bf"]["depth_acc"], floor=1e-2)
    eT = np.abs(b["fT"] - b["bf"]["final_T"])
    # the same deterministic bound at every size, 4K included: K1 carries the screen position in compensated
    # arithmetic relative to the Gaussian's tile rect (project.hip), so the 2.4e-4 px ulp of an absolute 4K
    # coordinate no longer reaches the image (round 2: 99.8 % of the 4K pixels within 1e-4, q99.9 = 1.2e-4;
    # now q99.99 = 3e-6).  Depth and final T hold 1e-4 everywhere; for RGB a handful of pixels (2 of 8 M at 4K)
    # sit just above it (1.9e-4), asserted as a count.
    assert ed[clear].max() < TOL and eT[clear].max() < TOL, (b["name"], ed[clear].max(), eT[clear].max())
    n_rgb = int((er[clear].max(-1) >= TOL).sum())
    assert n_rgb <= 4 and er[clear].max() < 5e-4, (b["name"], n_rgb, er[clear].max())
    assert np.quantile(er[clear], 0.9999) < 2e-5, (b["name"], np.quantile(er[clear], 0.9999))
    assert (b["last_hip"][clear] == b["last_ref"][clear]).mean() > 1.0 - 1e-12
    # decision-ambiguous pixels may flip one alpha_min / T_stop level contribution
    assert np.abs(b["rgb"] - b["bf"]["rgb"]).max() < 0.02
    # radius / visibility: integer decisions agree except within fp32 noise of an integer boundary
    f = splat_fields(b["sp"], b["radii"])
    same = f["radius"].numpy() == b["pc"]["radius"]
    assert same.mean() > 0.995


def test_tight_rect_drops_only_invisible_pairs(both):
    """Every pair in the normative B.4 rect but outside the product's tight rect has
    o*exp(-sigma) < 1/255 at every pixel centre of its tile (fp64 oracle values)."""
    b = both
    f = splat_fields(b["sp"], b["radii"])
    pc = b["pc"]
    mx, dropped = b["dropped_max_alpha"], b["dropped"]      # (computed in build_case: R.dropped_pairs_max_alpha)
    # the product may only drop pairs (subset), and it drops a substantial share on these scenes.
    # `same`: Gaussians whose NORMATIVE B.4 rect is the same in fp32 and fp64 (radius = ceil() of a value
    # within rounding of an integer, or (u -+ r)/16 within rounding of a tile boundary, differ in < 0.5 %)
    class _C:   # rect_from only needs .tiles
        tiles = ((b["W"] + 15) // 16, (b["H"] + 15) // 16)
    norm32 = rect_from(f["xy"], f["radius"], _C).numpy()
    same = (f["radius"].numpy() == pc["radius"]) & ((norm32 == pc["rect"]).all(axis=1) | (pc["radius"] == 0))
    assert same.mean() > 0.995
    assert dropped >= b["n_ref"] - b["n_index"] - 64 * int((~same).sum())
    assert dropped > 0.15 * b["n_ref"], (dropped, b["n_ref"])
    # the claim is about the tightening, i.e. about the Gaussians with an unambiguous normative rect
    # ... up to fp32 rounding of the projected covariance: for a Gaussian whose 3-D extent along the view
    # direction is orders of magnitude larger than its footprint the entries of J W Sigma W^T J^T are small
    # differences of large products, and the fp32 extent sqrt(tau^2 Sigma') can fall short of the fp64 one by
    # more than the 0.02 px margin (seen: 1 Gaussian of 5 M at 4K with alpha = 1.08 / 255 on one dropped tile)
    n_over = int((mx[same] >= 1.0 / 255.0).sum())
    assert n_over <= max(1, int(1e-6 * b["N"])) and mx[same].max() < 1.5 / 255.0, (b["name"], mx[same].max() * 255.0, n_over)
    assert int((mx[~same] >= 1.0 / 255.0).sum()) <= max(8, int(2e-5 * b["N"]))


def test_unlisted_pairs_are_invisible(both):
    """Pair by pair: every (tile, Gaussian) pair of the oracle's NORMATIVE lists that the product's lists lack (the
    tight rect of K1, or an fp32 normative rect one tile smaller) has o*exp(-sigma) < 1/255 at every pixel centre of
    its tile (fp64 oracle values) -- B.6 skips such pixels, so neither the image nor a gradient can change.  (A finer
    rule -- list a pair only if the ellipse {alpha >= 1/255} reaches the tile, VERDICT r3 4c -- was built and
    measured in round 4: -10 % pairs, K6 -1.3 %, K7 -0.2 %, K8 +14 %; not kept, profiles/r4_ab_runs.txt.)"""
    b = both
    ma = b["miss_alpha"]
    assert len(ma) >= b["n_ref"] - b["n_hip"]          # (equal unless the product lists hold a few pairs of their own)
    print(f"{b['name']}: normative pairs {b['n_ref']}, listed {b['n_hip']}; largest alpha*255 among the unlisted pairs {ma.max() * 255:.4f}")
    # same exceptions as the tight-rect test (fp32 extent of a near-degenerate projected covariance): counted, bounded
    f = splat_fields(b["sp"], b["radii"])
    same = (f["radius"].numpy() == b["pc"]["radius"])[b["miss_gid"]]
    assert int(((ma >= 1.0 / 255.0) & same).sum()) <= max(1, int(1e-6 * b["N"])), b["name"]
    assert int((ma >= 1.0 / 255.0).sum()) <= max(8, int(2e-5 * b["N"])), b["name"]
    assert ma[same].max() < 1.5 / 255.0, ma[same].max() * 255


GMARGIN = 1e-4     # decision margin below which a pixel counts as ambiguous for the GRADIENT classification
GTOL = 1e-4        # the bar: north_star / BASELINE.md "gradients <= 1e-4 relative"


def classify_gaussians(b, extra_unclear=None):
    """-> (pixel margin map, reaching, clear).  A pixel is decision-clear if every threshold test App. B.6
    evaluates for it (alpha against 1/255, T' against 1e-4) is further than GMARGIN (relative) from its
    threshold in the fp64 oracle and its tile's fp32 depth order equals the fp64 order.  A flipped decision at
    a pixel changes the transmittance of every later Gaussian there, so a GAUSSIAN is clear only if every
    pixel where it contributes or nearly contributes (alpha >= 0.5 / 255) is clear; `reaching` = has such a pixel."""
    pm = b["margin"].copy()
    pm[~b["order_ok"]] = 0.0
    if extra_unclear is not None:
        pm[extra_unclear] = 0.0
    pc = b["pc"]
    gmin, npix = b["R"].gaussian_min_margin(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], b["cb"],
                                            b["W"], b["H"], pm, near=0.5)
    reach = npix > 0
    return pm, reach, reach & (gmin > GMARGIN)


def max_over(n_clear):
    """Clear Gaussians allowed above the bar (and then below 10x the bar): 2 per million.  What they are (r4,
    tools/grad_offenders.py): (a) K7 forms the conic moments sum q (g - u)^2 from tile-centred sums (raster.hip),
    exact to ~1e-5 of (tile coordinate / footprint)^2 -- a Gaussian a pixel wide sitting 7 px off the tile centre
    reaches 3e-4 (1 of 2.2 M at 4K); (b) K8's quaternion gradient of a Gaussian whose three scales agree to 1e-5
    vanishes analytically, so |J| m is ~0 while the fp32 chain rounds at the size of its intermediates."""
    return max(1, int(2e-6 * n_clear))


def assert_gradients(tag, got, ref, mass, reach, clear, keys, report, tol_max=GTOL, tol_q=3e-5):
    """The gradient bar: on every clear Gaussian  max_c |HIP - oracle| <= tol_max * max_c (un-cancelled magnitude)
    for each gradient, up to max_over() named exceptions that stay below 10 tol_max, and q99.99 <= tol_q; the
    unclear ones are counted and bounded; Gaussians that reach no pixel have exactly zero gradients on both sides."""
    from tests.util import err_over_mass
    for key in keys:
        e = err_over_mass(got[key], ref[key], mass[key])
        ec, eu = e[clear], e[reach & ~clear]
        report.append(f"{key}: clear max {ec.max():.1e} q99.99 {np.quantile(ec, 0.9999):.1e} | unclear >{GTOL:g}: "
                      f"{int((eu > GTOL).sum())} of {eu.size}, max {eu.max() if eu.size else 0:.1e}")
        n_over = int((ec > tol_max).sum())
        assert n_over <= max_over(ec.size) and ec.max() <= 10 * tol_max and np.quantile(ec, 0.9999) <= tol_q, (
            tag, key, float(ec.max()), float(np.quantile(ec, 0.9999)), n_over, int(clear.sum()))
        # an ambiguous pixel flips at most contributions of weight ~1/255 (or the tail behind T = 1e-4): most unclear
        # Gaussians still agree; a flip can change a Gaussian whose gradient comes from that one pixel by O(1)
        assert (eu > GTOL).mean() < 0.02 and (eu.max() if eu.size else 0) < 4.0, (tag, key, float((eu > GTOL).mean()))
        g = np.asarray(got[key], np.float64).reshape(len(reach), -1)
        r = np.asarray(ref[key], np.float64).reshape(len(reach), -1)
        if key != "v_sh" and key != "v_means":   # (colour / view-direction gradients exist for unseen Gaussians too)
            assert np.abs(g[~reach]).max() <= 1e-30 + 1e-3 * np.abs(r[~reach]).max(), (tag, key)


def test_gradients_match_oracle_fullsize(both):
    """All gradients of the raster backward + projection backward against the fp64 C oracle at full size, random
    upstream gradients.  (1) K7's output -- the ten screen-space gradients per Gaussian (tgs_reduce_partials of the
    partial records) -- and (2) the five parameter gradients, each asserted at  max <= 1e-4  on every
    decision-clear Gaussian, relative to the un-cancelled magnitude of the gradient (oracle: the same walk with
    absolute values; for the parameters |J| times it, J = the oracle's projection backward column by column).
    Why that scale: with random upstream gradients a Gaussian's ~200 signed pixel contributions cancel to 1e-3 of
    their magnitude for one Gaussian in a thousand; no fp32 evaluation (the scalar fp32 build of the oracle
    included: tools/grad_err_k7.py) is accurate relative to such a remainder."""
    from tests.util import K7_KEYS, PARAM_KEYS, k7_outputs, param_mass
    b = both
    pm, reach, clear = classify_gaussians(b)
    report = [f"{b['name']}: decision-clear pixels {np.mean(pm > GMARGIN):.5f} (forward test, margin 1e-3: "
              f"{np.mean((b['margin'] > 1e-3) & b['order_ok']):.4f}); Gaussians reaching a pixel {reach.mean():.4f}, "
              f"of which clear {clear.sum() / max(reach.sum(), 1):.4f}"]
    frac = clear.sum() / max(reach.sum(), 1)
    assert abs(frac - CLEAR_GAUSSIANS_MEASURED[b["name"]]) < 0.02, report
    N = b["N"]
    ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
    assert_gradients(b["name"], k7_outputs(b["v_splats"]), ref7, b["m7"], reach, clear, K7_KEYS, report)
    pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], b["W"], b["H"], b["pc"]["radius"], b["m7"])
    got = dict(zip(PARAM_KEYS, b["grads"]))
    assert_gradients(b["name"], got, b["pb"], pmass, reach, clear, PARAM_KEYS, report)
    print("\n".join(report))
    _write_report(report)
    b["cache"]["grad_report"] = report
    # whole-tensor agreement (dominated by the unclear Gaussians)
    for key in PARAM_KEYS:
        ref, g = b["pb"][key].reshape(N, -1), np.asarray(got[key]).reshape(N, -1)
        cos = (ref * g).sum() / np.sqrt((ref * ref).sum() * (g * g).sum())
        rel_l2 = np.sqrt(((g - ref) ** 2).sum() / (ref * ref).sum())
        assert cos > 0.9999 and rel_l2 < 1e-2, (b["name"], key, cos, rel_l2)


# ---------------------------------------------------------------------------------------------
# the loss side of the step at full size
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("W,H", [(800, 800), (1920, 1080), (3840, 2160)])
def test_ssim_matches_oracle_fullsize(dev, W, H):
    """K10 at the BASELINE image sizes (the segment length of k_ssim_fwd / k_ssim_bwd depends on the image
    size: imgloss.hip picks 12 / 34 / 45-row segments here) against the fp64 torch SSIM and its autograd."""
    from oracle import torch_oracle as O
    from touch_gs_amd import ops
    g = torch.Generator().manual_seed(W + H)
    # a smooth image + noise: SSIM well inside (0, 1), non-trivial local statistics
    base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 40 + 2, W // 40 + 2, generator=g, dtype=torch.float64),
                                           size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    gt = base.clamp(0, 1).contiguous()
    img = (base + 0.1 * torch.randn(H, W, 3, generator=g, dtype=torch.float64)).clamp(0, 1).contiguous()
    tot, v = ops.ssim_fwd_bwd(img.float().to(dev), gt.float().to(dev), weight=-0.2 / (3 * H * W))
    if H <= 1080:
        img.requires_grad_(True)
        s = O.ssim(img, gt)
        (0.2 * (1 - s)).backward()
        ref = img.grad.numpy()
        got = v.cpu().double().numpy()
    else:
        # 4K: the fp64 autograd of the whole image costs a minute of the GPU-test budget.  SSIM is local -- the gradient at a
        # pixel depends on the 21 x 21 pixels around it -- so the oracle differentiates three full-width bands (top edge,
        # middle, bottom edge; every strip of the kernels, both image borders, segment boundaries inside) and the
        # comparison keeps the rows at least 10 pixels away from a cut; the map sum is checked on the whole image (no grad).
        with torch.no_grad():
            s = O.ssim(img, gt)
        refs, gots = [], []
        for a, b in ((0, 170), (H // 2 - 85, H // 2 + 85), (H - 170, H)):
            crop = img[a:b].clone().requires_grad_(True)
            (0.2 * (1 - O.ssim(crop, gt[a:b])) * ((b - a) / H)).backward()      # mean over the band -> share of the image mean
            lo, hi = (0 if a == 0 else 10), ((b - a) if b == H else (b - a) - 10)
            refs.append(crop.grad[lo:hi].numpy())
            gots.append(v[a + lo:a + hi].cpu().double().numpy())
        ref, got = np.concatenate(refs), np.concatenate(gots)
    assert abs(tot.item() / (3 * H * W) - s.item()) < 1e-5, (tot.item() / (3 * H * W), s.item())
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() < 1e-4 * scale, np.abs(got - ref).max() / scale
    rel_l2 = np.sqrt(((got - ref) ** 2).sum() / (ref * ref).sum())
    assert rel_l2 < 5e-5, rel_l2          # measured 1.05e-5 .. 1.09e-5 at the three sizes (fp32 moments)


# the reference's three flag sets: scripts/train_bunny_real.sh:52, train_block_data.sh:50 (= train_mirror.sh:49),
# train_bunny_blender.sh:50
FLAG_SETS = {"bunny_real": ("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 0.005, 0.01),
             "block": ("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 0.2, 1.0),
             "bunny_blender": ("SIMPLE_LOSS", 0.5, 1.0)}


def test_train_step_matches_oracle_fullsize(both, dev):
    """The fused train step (K1..K8 + K10: L1 + SSIM + tactile depth / uncertainty loss evaluated inside
    K7) at configs[1] and configs[2]: loss terms and all five parameter gradients against
    O.train_loss (fp64 torch) differentiated through the fp64 C oracle's compositing / projection
    backward, for the reference's three flag sets."""
    from oracle import torch_oracle as O
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig, View
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view
    # the per-GPU part of configs[4] (5 M, 4K) runs ONE flag set (round 6; rounds 3-5 skipped it: the budget is there)
    _train_step_case(both, dev, FLAG_SETS if both["W"] <= 2048 else {"block": FLAG_SETS["block"]})


@pytest.mark.parametrize("name", ["clamp_640", "hard_640"])
def test_train_step_on_the_clamp_and_on_hard_strata(dev, name):
    """The same fused train step against the same oracle chain on the two scenes of the strict-bar tests below (pixels on
    the alpha = 0.999 clamp; needles, sub-pixel, near, loud-SH and far strata), with the flag set that has every term on
    (depth loss with the uncertainty weighting)."""
    _train_step_case(build_case(name, dev), dev, {"block": FLAG_SETS["block"]})


def _train_step_case(b, dev, flag_sets):
    from oracle import torch_oracle as O
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig, View
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view
    from tests.util import PARAM_KEYS, param_mass
    N, W, H, deg = b["N"], b["W"], b["H"], b["deg"]
    mv = make_view(N, W, H, deg, b["seed"], dev, view=b["view"], n_views=8)
    view = View(cam=b["cam"], rgb=mv.rgb, depth=mv.depth, uncertainty=mv.uncertainty)
    gt64, dgt64, unc64 = (t.cpu().double() for t in (mv.rgb, mv.depth, mv.uncertainty))
    R, cb, pc, bf, Pn = b["R"], b["cb"], b["pc"], b["bf"], b["Pn"]
    for tag, (ltype, mult, uw) in flag_sets.items():
        # ---- HIP: the fused step's forward + backward ----
        D = b["D"]
        params = GaussianParams.from_tensors(*[D[k].clone() for k in GaussianParams.NAMES])
        cfg = ModelConfig(sh_degree=deg, sh_degree_interval=0, depth_loss_mult=mult, depth_loss_type=ltype,
                          uncertainty_weight=uw)
        model = DepthGaussianSplattingModel(cfg, params)
        tile_loss, ssim_sum = model.forward_backward(view)
        losses = {k: float(v) for k, v in model.loss_from(tile_loss, ssim_sum, view).items()}
        ssim_hip = float(ssim_sum.sum()) / (3 * H * W)
        got = {k: params.g[k].detach().cpu().double().numpy() for k in GaussianParams.NAMES}
        # ---- oracle: fp64 loss on the C oracle's images, its autograd seeds the C oracle's backward ----
        rgb_t = torch.from_numpy(bf["rgb"]).requires_grad_(True)
        dacc_t = torch.from_numpy(bf["depth_acc"]).requires_grad_(True)
        alpha_t = (1.0 - torch.from_numpy(bf["final_T"])).requires_grad_(True)
        out = dict(rgb=rgb_t, depth_acc=dacc_t, alpha=alpha_t)
        L = O.train_loss(out, gt64, dgt64, unc64, ssim_lambda=0.2, depth_loss_mult=mult, depth_loss_type=ltype,
                         uncertainty_weight=uw)
        L.backward()
        L_depth = mult * float(O.depth_loss(dacc_t.detach(), alpha_t.detach(), dgt64, unc64, ltype, uw))
        ssim_ref = float(O.ssim(rgb_t.detach(), gt64))
        bb = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], b["g2"], b["ts2"], cb, W, H,
                         bf["final_T"], bf["final_idx"], rgb_t.grad.numpy(), dacc_t.grad.numpy(), alpha_t.grad.numpy())
        pb = R.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H,
                           pc["radius"], bb["v_xy"], bb["v_conic"], bb["v_opac"], bb["v_rgb"], bb["v_depth"])
        # ---- loss terms ----
        total = sum(losses.values())
        assert abs(total - float(L.detach())) < 2e-5 * abs(float(L.detach())), (b["name"], tag, total, float(L.detach()))
        assert abs(losses["depth_loss"] - L_depth) < 1e-4 * abs(L_depth) + 1e-9, (b["name"], tag, losses["depth_loss"], L_depth)
        assert abs(ssim_hip - ssim_ref) < 1e-5, (b["name"], tag, ssim_hip, ssim_ref)
        # ---- gradients: max <= 1e-4 of the un-cancelled magnitude on every decision-clear Gaussian ----
        # (besides the compositing thresholds, the L1 term's sign(C - gt) is a decision: pixels with a channel
        # closer than 2e-4 -- the forward's own tolerance -- to the ground truth count as ambiguous)
        l1_amb = (np.abs(bf["rgb"] - gt64.numpy()) < 2e-4).any(-1)
        pm, reach, clear = classify_gaussians(b, extra_unclear=l1_amb)
        m7 = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], b["g2"], b["ts2"], cb, W, H,
                         bf["final_T"], bf["final_idx"], rgb_t.grad.numpy(), dacc_t.grad.numpy(), alpha_t.grad.numpy(),
                         mass=True)
        pmass = param_mass(R, Pn, deg, cb, W, H, pc["radius"], m7)
        report = [f"{b['name']} / {tag}: decision-clear pixels {np.mean(pm > GMARGIN):.5f}, clear Gaussians "
                  f"{clear.sum() / max(reach.sum(), 1):.4f} of the reaching ones"]
        want = CLEAR_GAUSSIANS_MEASURED_TRAIN.get(b["name"])
        frac = clear.sum() / max(reach.sum(), 1)
        assert want is None or abs(frac - want) < 0.02, report
        gotk = {"v_" + k: got[k] for k in GaussianParams.NAMES}
        # Here the upstream gradient images are themselves computed twice -- fp32 kernels on the HIP image, fp64
        # torch on the oracle's image -- and the loss curvature (SSIM: 1 / (sigma^2 + C2), C2 = 9e-4) amplifies the
        # 1e-4 the images may differ by: 99.99 % of the clear Gaussians within 1e-4, every one within 1e-3.  The
        # bar proper (max <= 1e-4 with the upstream gradients given) is test_gradients_match_oracle_fullsize.
        assert_gradients(f"{b['name']}/{tag}", gotk, pb, pmass, reach, clear, PARAM_KEYS, report, tol_max=1e-3, tol_q=GTOL)
        print("\n".join(report))
        _write_report(report)
        for name, key in (("means", "v_means"), ("log_scales", "v_log_scales"), ("quats", "v_quats"),
                          ("opac_logit", "v_opac_logit"), ("sh", "v_sh")):
            ref = pb[key].reshape(got[name].shape)
            cos = (ref * got[name]).sum() / np.sqrt((ref * ref).sum() * (got[name] * got[name]).sum())
            rel_l2 = np.sqrt(((got[name] - ref) ** 2).sum() / (ref * ref).sum())
            assert cos > 0.9999 and rel_l2 < 1e-2, (b["name"], tag, name, cos, rel_l2)



def test_gradients_with_screen_filling_gaussians(dev):
    """K8's segmented sum on LONG runs (round 6) and the front half on huge rects, against the fp64 oracle: a 720p frame
    with a dozen Gaussians that cover 2000 - 3600 tiles each (what the table / background Gaussians of a converged
    object-centric model do -- profiles/r6_before_ckpt_loop_*.json: max_hits 2695 - 3600).  Until round 6 one thread
    added such a run record by record; now the workgroup's 16-lane teams sum it in 64-record segments (project.hip,
    group_sum_partials).  Same bar as the full-size test for everything decision-clear; the huge Gaussians reach every
    pixel of the image, so some pixel of theirs is always decision-ambiguous and the clear / unclear classification says
    nothing about them -- but a flipped 1/255 contribution is nothing against the sum over 900 k pixels, so THEIR ten
    screen-space gradients and five parameter gradients are asserted directly, at 1e-4 of the un-cancelled magnitude."""
    from tests.util import K7_KEYS, PARAM_KEYS, err_over_mass, k7_outputs, param_mass
    b = build_case("huge_720p", dev)
    N = b["N"]
    f = splat_fields(b["sp"], b["radii"])
    hits = f["hits"].numpy()
    huge = np.zeros(N, bool)
    huge[::HUGE_EVERY] = True
    assert (hits[huge] >= 2000).sum() >= 8, hits[huge]
    pm, reach, clear = classify_gaussians(b)
    report = [f"huge_720p: {int(huge.sum())} blown-up Gaussians with {hits[huge].min()} - {hits[huge].max()} tiles, "
              f"{b['n_hip']} pairs; clear Gaussians {clear.sum() / max(reach.sum(), 1):.4f} of the reaching ones"]
    ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
    got7 = k7_outputs(b["v_splats"])
    assert_gradients("huge_720p", got7, ref7, b["m7"], reach, clear, K7_KEYS, report)
    pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], b["W"], b["H"], b["pc"]["radius"], b["m7"])
    got = dict(zip(PARAM_KEYS, b["grads"]))
    assert_gradients("huge_720p", got, b["pb"], pmass, reach, clear, PARAM_KEYS, report)
    for keys, g_, r_, m_ in ((K7_KEYS, got7, ref7, b["m7"]), (PARAM_KEYS, got, b["pb"], pmass)):
        for key in keys:
            e = err_over_mass(g_[key], r_[key], m_[key])[huge & (hits >= 2000)]
            report.append(f"{key}: the huge Gaussians' error / mass: max {e.max():.1e}")
            assert e.max() < GTOL, (key, e)
    print("\n".join(report))


@pytest.mark.parametrize("quad", [False, True])
def test_gradients_on_the_alpha_clamp_in_both_forms_of_k7(dev, quad):
    """The full-size test's bar -- max <= 1e-4 of the un-cancelled magnitude on EVERY decision-clear Gaussian, ten
    screen-space and five parameter gradients -- on a scene whose pixels sit on the alpha = 0.999 clamp by the ten
    thousand (the BASELINE scenes have almost none), through K7's one-wave form and through its four-wave form (rule
    forced onto every tile walking more than 8 entries).  The small-size backward test (tests/test_gpu_parity.py::
    test_rasterize_bwd) asserts a share of bad entries and a median; this one has no statistical escape."""
    from tests.util import K7_KEYS, PARAM_KEYS, k7_outputs, param_mass
    from touch_gs_amd import ops
    before = ops.set_k7_quad()
    try:
        ops.set_k7_quad(*((1, 8) if quad else (0, before[1])))
        b = build_case("clamp_640", dev)
    finally:
        ops.set_k7_quad(*before)
    N, W, H = b["N"], b["W"], b["H"]
    pc = b["pc"]
    # pairs on the clamp, counted with the oracle's own projection: o exp(-sigma) > 0.999 at a pixel centre
    n_cl = 0
    for g in np.nonzero((pc["opac"] > 0.999) & (pc["tiles_hit"] > 0))[0]:      # (the region is a fraction of the footprint)
        x0, y0 = int(min(max(pc["xy"][g, 0] - 4, 0), W)), int(min(max(pc["xy"][g, 1] - 4, 0), H))
        ys, xs = np.mgrid[y0:min(y0 + 8, H), x0:min(x0 + 8, W)]
        dx, dy = pc["xy"][g, 0] - (xs + 0.5), pc["xy"][g, 1] - (ys + 0.5)
        sig = 0.5 * (pc["conic"][g, 0] * dx * dx + pc["conic"][g, 2] * dy * dy) + pc["conic"][g, 1] * dx * dy
        n_cl += int((pc["opac"][g] * np.exp(-sig) > 0.999).sum())
    assert n_cl >= 3000, n_cl
    pm, reach, clear = classify_gaussians(b)
    frac = clear.sum() / max(reach.sum(), 1)
    report = [f"clamp_640 (four-wave form: {quad}): >= {n_cl} pixel-Gaussian pairs on the clamp; "
              f"clear Gaussians {frac:.4f} of the reaching ones"]
    assert frac > 0.25, report
    ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
    assert_gradients("clamp_640", k7_outputs(b["v_splats"]), ref7, b["m7"], reach, clear, K7_KEYS, report)
    pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], W, H, pc["radius"], b["m7"])
    assert_gradients("clamp_640", dict(zip(PARAM_KEYS, b["grads"])), b["pb"], pmass, reach, clear, PARAM_KEYS, report)
    print("\n".join(report))
    _write_report(report)
    # the two forms did take different paths: same gradients up to rounding, not the same bits
    _CLAMP_FORMS[quad] = b["v_splats"]
    if len(_CLAMP_FORMS) == 2:
        a, c = _CLAMP_FORMS[False], _CLAMP_FORMS[True]
        assert not np.array_equal(a, c) and np.abs(a - c).max() <= 1e-3 * np.abs(a).max()


_CLAMP_FORMS = {}


def test_gradients_on_hard_strata(dev):
    """The full-size test's bar on the strata of hard_640 (see EXTRA_CONFIGS): forward at 1e-4 on every decision-clear
    pixel, ten screen-space and five parameter gradients at 1e-4 of the un-cancelled magnitude on every decision-clear
    Gaussian, stratum by stratum in the report."""
    from tests.util import K7_KEYS, PARAM_KEYS, err_over_mass, k7_outputs, param_mass
    b = build_case("hard_640", dev)
    N, W, H = b["N"], b["W"], b["H"]
    clear_px = (b["margin"] > 1e-3) & b["order_ok"]
    assert clear_px.mean() > 0.85, clear_px.mean()
    er = relerr(b["rgb"], b["bf"]["rgb"], floor=1e-2)[clear_px]
    ed = relerr(b["depth"], b["bf"]["depth_acc"], floor=1e-2)[clear_px]
    eT = np.abs(b["fT"] - b["bf"]["final_T"])[clear_px]
    assert er.max() <= TOL and ed.max() <= TOL and eT.max() <= TOL, (er.max(), ed.max(), eT.max())
    pm, reach, clear = classify_gaussians(b)
    names = ("needle", "sub-pixel", "near", "faint", "loud SH", "far", "plain", "plain")
    report = [f"hard_640: decision-clear pixels {clear_px.mean():.4f}; rgb / depth / T max {er.max():.1e} {ed.max():.1e} {eT.max():.1e}; "
              + ", ".join(f"{names[k]}: {int((reach & (np.arange(N) % 8 == k)).sum())} reaching / {int((clear & (np.arange(N) % 8 == k)).sum())} clear"
                          for k in range(6))]
    assert int((reach & (np.arange(N) % 8 == 3)).sum()) == 0          # the faint stratum reaches no pixel
    for k in (0, 1, 2, 4, 5):
        assert int((clear & (np.arange(N) % 8 == k)).sum()) >= 150, report
    ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
    got7 = k7_outputs(b["v_splats"])
    # (tol_q = the bar itself: with 17 k Gaussians q99.99 is the second largest value)
    assert_gradients("hard_640", got7, ref7, b["m7"], reach, clear, K7_KEYS, report, tol_q=GTOL)
    pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], W, H, b["pc"]["radius"], b["m7"])
    got = dict(zip(PARAM_KEYS, b["grads"]))
    assert_gradients("hard_640", got, b["pb"], pmass, reach, clear, PARAM_KEYS, report, tol_q=GTOL)
    for k in (0, 1, 2, 4, 5):
        sel = clear & (np.arange(N) % 8 == k)
        worst = max(float(err_over_mass(g_[key], r_[key], m_[key])[sel].max())
                    for keys, g_, r_, m_ in ((K7_KEYS, got7, ref7, b["m7"]), (PARAM_KEYS, got, b["pb"], pmass)) for key in keys)
        report.append(f"{names[k]}: worst error / mass over the fifteen gradients {worst:.1e}")
    print("\n".join(report))
    _write_report(report)


def test_forward_and_gradients_with_an_off_centre_camera(dev):
    """fx != fy and a principal point away from the image centre (every other scene of the suite is symmetric): forward at
    1e-4 on the decision-clear pixels, all fifteen gradients at 1e-4 of the un-cancelled magnitude on the decision-clear
    Gaussians, and the fused train step through the same oracle chain."""
    from tests.util import K7_KEYS, PARAM_KEYS, k7_outputs, param_mass
    b = build_case("skewed_640", dev)
    N, W, H = b["N"], b["W"], b["H"]
    assert abs(b["cam"].fx / b["cam"].fy - 1.15) < 1e-6 and abs(b["cam"].cx - (W / 2 + 37.0)) < 1e-6
    clear_px = (b["margin"] > 1e-3) & b["order_ok"]
    assert clear_px.mean() > 0.9, clear_px.mean()
    er = relerr(b["rgb"], b["bf"]["rgb"], floor=1e-2)[clear_px]
    ed = relerr(b["depth"], b["bf"]["depth_acc"], floor=1e-2)[clear_px]
    eT = np.abs(b["fT"] - b["bf"]["final_T"])[clear_px]
    assert er.max() <= TOL and ed.max() <= TOL and eT.max() <= TOL, (er.max(), ed.max(), eT.max())
    assert np.array_equal(b["last_hip"][clear_px], b["last_ref"][clear_px])
    pm, reach, clear = classify_gaussians(b)
    report = [f"skewed_640: decision-clear pixels {clear_px.mean():.4f}; rgb / depth / T max {er.max():.1e} {ed.max():.1e} {eT.max():.1e}; "
              f"clear Gaussians {clear.sum() / max(reach.sum(), 1):.4f} of the reaching ones"]
    ref7 = {k: np.asarray(b["bb"][k], np.float64).reshape(N, -1) for k in K7_KEYS}
    assert_gradients("skewed_640", k7_outputs(b["v_splats"]), ref7, b["m7"], reach, clear, K7_KEYS, report, tol_q=GTOL)
    pmass = param_mass(b["R"], b["Pn"], b["deg"], b["cb"], W, H, b["pc"]["radius"], b["m7"])
    assert_gradients("skewed_640", dict(zip(PARAM_KEYS, b["grads"])), b["pb"], pmass, reach, clear, PARAM_KEYS, report, tol_q=GTOL)
    print("\n".join(report))
    _write_report(report)
    _train_step_case(b, dev, {"block": FLAG_SETS["block"]})

