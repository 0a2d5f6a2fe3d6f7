"""Full-size oracle parity on the BASELINE configs: the HIP path against the
fp64 build of the scalar C oracle (oracle/ref_raster.c) on

* configs[1]: 100 k Gaussians, 800x800   (seed 1235),
* configs[2]: 1 M Gaussians, 1920x1080   (seed 1236),
* configs[4], the part one GPU runs: 5 M Gaussians, SH degree 3, 3840x2160 (seed 1238),

forward (RGB / depth / final T / last contributor) on every pixel whose threshold decisions are not
ambiguous under fp32 rounding, all five parameter gradients, and the output-preservation claim of
the tight tile rectangle: every (tile, Gaussian) pair that the product drops
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


@pytest.fixture(scope="module", params=list(CONFIGS))
def both(request, dev):
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg, seed, view = CONFIGS[request.param]
    P, intr = synthetic_gaussians(N, W, H, deg, seed)
    cam = make_camera(intr, view, 8, bg=(0.1, 0.2, 0.3))
    D = {k: v.to(dev).contiguous() for k, v in P.items()}
    # ---- HIP ----
    sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                     D["sh"], deg, want_radii=True)
    n_hip = st.tolist()[0]
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
    cb = R.cam_block(np.asarray(cam.viewmat, np.float64).reshape(4, 4), cam.fx, cam.fy, cam.cx, cam.cy, bg=cam.bg)
    Pn = {k: n64(v) for k, v in P.items()}
    pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
    g2, ts2 = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
    bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
    margin = R.blend_margin(pc["xy"], pc["conic"], pc["opac"], g2, ts2, cb, W, H)
    bb = R.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H,
                     bf["final_T"], bf["final_idx"], n64(v_rgb), n64(v_d), n64(v_a))
    pb = R.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H,
                       pc["radius"], bb["v_xy"], bb["v_conic"], bb["v_opac"], bb["v_rgb"], bb["v_depth"])
    tile_np = tile.cpu().numpy()
    pos2 = ts2[tile_np] + np.maximum(bf["final_idx"], 0)
    last_gid_ref = np.where(bf["final_idx"] >= 0, g2[np.minimum(pos2, max(len(g2) - 1, 0))], -1)
    return dict(name=request.param, N=N, W=W, H=H, R=R, cb=cb, sp=sp, radii=radii, n_hip=n_hip, n_ref=len(g2),
                rgb=rgb.cpu().numpy(), depth=depth.cpu().numpy(), fT=fT.cpu().numpy(), last_hip=last_gid_hip,
                grads=[t.cpu().double().numpy() for t in grads[:5]], pc=pc, bf=bf, margin=margin, pb=pb,
                last_ref=last_gid_ref)


def test_forward_matches_oracle_fullsize(both):
    b = both
    clear = b["margin"] > 1e-3
    assert clear.mean() > 0.9, clear.mean()
    er = relerr(b["rgb"], b["bf"]["rgb"], floor=1e-2)
    ed = relerr(b["depth"], b["bf"]["depth_acc"], floor=1e-2)
    eT = np.abs(b["fT"] - b["bf"]["final_T"])
    if b["W"] <= 2048:
        assert er[clear].max() < TOL and ed[clear].max() < TOL and eT[clear].max() < TOL, \
            (b["name"], er[clear].max(), ed[clear].max(), eT[clear].max())
    else:
        # 4K: pixel coordinates up to 3840 carry an fp32 ulp of 2.4e-4 px, i.e. up to ~1e-3 relative on
        # the alpha of a sub-pixel Gaussian -- the sum of ~100 contributions per pixel sits AT the 1e-4
        # bar (measured: 99.8 % of the clear pixels within 1e-4, q99.9 = 1.2e-4, independent of the
        # decision margin).  Any fp32 rasterizer shares this floor; the bound here is statistical.
        for e, name in ((er, "rgb"), (ed, "depth"), (eT[..., None], "final_T")):
            v = e[clear]
            assert (v < TOL).mean() > 0.995, (b["name"], name, (v < TOL).mean())
            assert np.quantile(v, 0.9999) < 5e-4, (b["name"], name, np.quantile(v, 0.9999))
    assert (b["last_hip"][clear] == b["last_ref"][clear]).mean() > (0.9999 if b["W"] > 2048 else 1.0 - 1e-12)
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
    tight = f["rect"].numpy().astype(np.int32)
    tight[(f["hits"] == 0).numpy()] = 0
    pc = b["pc"]
    mx, dropped = b["R"].dropped_pairs_max_alpha(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], tight,
                                                 b["cb"], b["W"], b["H"])
    # the product may only drop pairs (subset), and it drops a substantial share on these scenes.
    # `same`: Gaussians whose NORMATIVE B.4 rect is the same in fp32 and fp64 (radius = ceil() of a value
    # within rounding of an integer, or (u -+ r)/16 within rounding of a tile boundary, differ in < 0.5 %)
    class _C:   # rect_from only needs .tiles
        tiles = ((b["W"] + 15) // 16, (b["H"] + 15) // 16)
    norm32 = rect_from(f["xy"], f["radius"], _C).numpy()
    same = (f["radius"].numpy() == pc["radius"]) & ((norm32 == pc["rect"]).all(axis=1) | (pc["radius"] == 0))
    assert same.mean() > 0.995
    assert dropped >= b["n_ref"] - b["n_hip"] - 64 * int((~same).sum())
    assert dropped > 0.15 * b["n_ref"], (dropped, b["n_ref"])
    # the claim is about the tightening, i.e. about the Gaussians with an unambiguous normative rect
    assert mx[same].max() < 1.0 / 255.0, (b["name"], mx[same].max() * 255.0, int((mx[same] >= 1.0 / 255.0).sum()))
    assert int((mx[~same] >= 1.0 / 255.0).sum()) <= max(8, int(2e-5 * b["N"]))


def test_gradients_match_oracle_fullsize(both):
    b = both
    names = ("means", "log_scales", "quats", "opac_logit", "sh")
    for name, got, key in zip(names, b["grads"], ("v_means", "v_log_scales", "v_quats", "v_opac_logit", "v_sh")):
        ref = b["pb"][key].reshape(got.shape)
        scale = np.abs(ref).max()
        e = relerr(got, ref, floor=1e-3 * scale)
        cos = (ref * got).sum() / np.sqrt((ref * ref).sum() * (got * got).sum())
        rel_l2 = np.sqrt(((got - ref) ** 2).sum() / (ref * ref).sum())
        assert np.median(e) < 3e-5, (b["name"], name, np.median(e))
        assert np.quantile(e, 0.98) < 5e-3, (b["name"], name, np.quantile(e, 0.98))
        assert cos > 0.9999 and rel_l2 < 1e-2, (b["name"], name, cos, rel_l2)
