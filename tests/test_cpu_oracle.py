"""CPU tests of the oracle itself: the two independent restatements agree, autograd matches
finite differences, and the committed self-oracle fixtures are reproduced."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from oracle.ref_c import RefC
from tests.util import clamp_scene, count_clamped, scene

GOLD = os.path.join(os.path.dirname(__file__), "golden", "raster_selforacle.npz")


@pytest.fixture(scope="module")
def refc():
    return RefC("f64")


@pytest.mark.parametrize("N,W,H,deg,seed,clamp", [(300, 80, 48, 3, 11, False), (150, 50, 35, 1, 12, False),
                                                   (60, 32, 32, 0, 13, False), (300, 80, 48, 3, 11, True),
                                                   (200, 64, 64, 2, 14, True)])
def test_c_and_torch_restatements_agree(refc, N, W, H, deg, seed, clamp):
    """The two independent restatements agree in value AND gradient -- also ON the alpha = 0.999
    clamp, where App. B.7 prescribes a pass-through gradient (clamp=True scenes hold > 50 clamped
    (pixel, Gaussian) pairs; torch.clamp's own autograd would disagree there by ~1e-3 relative)."""
    P, cam = (clamp_scene if clamp else scene)(N, W, H, deg, seed)
    for k in P:
        P[k].requires_grad_(True)
    out, pr, gid, ts = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, deg)
    n = lambda t: t.detach().numpy()
    if clamp:
        assert count_clamped(pr, gid, ts, cam) > 50
    cb = refc.cam_block(cam.viewmat.numpy(), cam.fx, cam.fy, cam.cx, cam.cy, bg=cam.bg)
    pc = refc.project_fwd(n(P["means"]), n(P["log_scales"]), n(P["quats"]), n(P["opac_logit"]), n(P["sh"]), deg, cb, W, H)
    v = pr["valid"].numpy()
    assert np.array_equal(n(pr["radius"]), pc["radius"]) and np.array_equal(n(pr["tiles_hit"]), pc["tiles_hit"])
    for k in ("xy", "conic"):
        assert np.abs(n(pr[k])[v] - pc[k][v]).max() < 1e-10
    for k in ("rgb", "opac", "depth"):
        assert np.abs(n(pr[k]) - pc[k]).max() < 1e-12
    g2, ts2 = refc.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
    assert np.array_equal(g2, gid) and np.array_equal(ts2, ts)
    bf = refc.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H)
    for k in ("rgb", "depth_acc", "final_T"):
        assert np.abs(n(out[k]) - bf[k]).max() < 1e-12
    assert np.array_equal(n(out["final_idx"]), bf["final_idx"])
    g = torch.Generator().manual_seed(seed)
    vr = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    vd = torch.randn(H, W, generator=g, dtype=torch.float64)
    va = torch.randn(H, W, generator=g, dtype=torch.float64)
    ((out["rgb"] * vr).sum() + (out["depth_acc"] * vd).sum() + (out["alpha"] * va).sum()).backward()
    bb = refc.blend_bwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g2, ts2, cb, W, H,
                        bf["final_T"], bf["final_idx"], n(vr), n(vd), n(va))
    pb = refc.project_bwd(n(P["means"]), n(P["log_scales"]), n(P["quats"]), n(P["opac_logit"]), n(P["sh"]), deg, cb,
                          W, H, pc["radius"], bb["v_xy"], bb["v_conic"], bb["v_opac"], bb["v_rgb"], bb["v_depth"])
    for k, kk in (("means", "v_means"), ("log_scales", "v_log_scales"), ("quats", "v_quats"),
                  ("opac_logit", "v_opac_logit"), ("sh", "v_sh")):
        ref = P[k].grad.numpy()
        assert np.abs(ref - pb[kk]).max() < 1e-9 * max(1.0, np.abs(ref).max()), k


def test_autograd_matches_finite_differences():
    P, cam = scene(120, 48, 32, 2, 5)
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(32, 48, 3, generator=g, dtype=torch.float64)
    dgt = torch.rand(32, 48, generator=g, dtype=torch.float64) * 5
    dgt[torch.rand(32, 48, generator=g) < 0.3] = 0
    unc = torch.rand(32, 48, generator=g, dtype=torch.float64) + 0.01

    def f():
        o, _, _, _ = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, 2)
        return O.train_loss(o, gt, dgt, unc, ssim_lambda=0.2, depth_loss_mult=0.2, uncertainty_weight=0.5)

    for k in P:
        P[k].requires_grad_(True)
    f().backward()
    with torch.no_grad():
        for name in P:
            gr = P[name].grad
            idx = gr.abs().flatten().argmax().item()
            flat = P[name].view(-1)
            old, h = flat[idx].item(), 1e-6
            flat[idx] = old + h
            fp = f().item()
            flat[idx] = old - h
            fm = f().item()
            flat[idx] = old
            fd = (fp - fm) / (2 * h)
            assert abs(fd - gr.flatten()[idx].item()) < 1e-5 * max(1.0, abs(fd)), name


def test_compositing_identities():
    """rgb = sum w c + T_final bg, alpha = 1 - T_final; zero-opacity Gaussians contribute nothing."""
    P, cam = scene(200, 64, 48, 1, 9, bg=(0.3, 0.6, 0.9))
    out, pr, gid, ts = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, 1)
    assert torch.allclose(out["alpha"], 1 - out["final_T"])
    P2 = {k: v.clone() for k, v in P.items()}
    P2["opac_logit"][::2] = -40.0
    out2, *_ = O.render(P2["means"], P2["log_scales"], P2["quats"], P2["opac_logit"], P2["sh"], cam, 1)
    keep = torch.arange(200)[1::2]
    P3 = {k: v[keep] for k, v in P.items()}
    out3, *_ = O.render(P3["means"], P3["log_scales"], P3["quats"], P3["opac_logit"], P3["sh"], cam, 1)
    assert torch.allclose(out2["rgb"], out3["rgb"], atol=1e-12)
    # black scene with bg: rgb == T * bg
    P4 = {k: v.clone() for k, v in P.items()}
    P4["sh"][:] = 0
    P4["sh"][:, 0, :] = -0.5 / O.SH_C0
    out4, *_ = O.render(P4["means"], P4["log_scales"], P4["quats"], P4["opac_logit"], P4["sh"], cam, 1)
    assert torch.allclose(out4["rgb"], out4["final_T"][..., None] * torch.tensor(cam.bg, dtype=torch.float64), atol=1e-12)


def test_selforacle_fixtures_reproduce():
    z = np.load(GOLD)
    for name in ("one", "seven", "two_hundred"):
        N, W, H, deg, seed = z[f"{name}/meta"]
        fx, fy, cx, cy = z[f"{name}/intr"]
        cam = O.Camera(viewmat=torch.from_numpy(z[f"{name}/viewmat"]), fx=fx, fy=fy, cx=cx, cy=cy, W=int(W), H=int(H),
                       bg=tuple(z[f"{name}/bg"]))
        P = {k: torch.from_numpy(z[f"{name}/in/{k}"]).requires_grad_(True) for k in
             ("means", "log_scales", "quats", "opac_logit", "sh")}
        out, _, gid, ts = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, int(deg))
        for k in ("rgb", "depth_acc", "alpha"):
            assert np.abs(out[k].detach().numpy() - z[f"{name}/out/{k}"]).max() < 1e-6
        assert np.array_equal(gid, z[f"{name}/sorted_gid"]) and np.array_equal(ts, z[f"{name}/tile_start"])
        L = (out["rgb"] * torch.from_numpy(z[f"{name}/w_rgb"])).sum() + \
            (out["depth_acc"] * torch.from_numpy(z[f"{name}/w_depth"])).sum() + \
            (out["alpha"] * torch.from_numpy(z[f"{name}/w_alpha"])).sum()
        L.backward()
        for k in P:
            assert np.allclose(P[k].grad.numpy(), z[f"{name}/grad/{k}"], rtol=1e-9, atol=1e-12)


def test_single_gaussian_closed_form():
    """One Gaussian, pixel at its centre: alpha = min(0.999, sigmoid(o)), rgb = alpha c + (1-alpha) bg."""
    z = np.load(GOLD)
    o = 1 / (1 + np.exp(-z["one/in/opac_logit"][0]))
    rgb, alpha = z["one/out/rgb"], z["one/out/alpha"]
    i = np.unravel_index(alpha.argmax(), alpha.shape)
    assert alpha[i] <= min(0.999, o) + 1e-6 and alpha[i] > 0.8 * min(0.999, o)
    c = np.maximum(z["one/in/sh"][0, 0] * O.SH_C0 + 0.5, 0)
    assert np.allclose(rgb[i], alpha[i] * c + (1 - alpha[i]) * z["one/bg"], atol=1e-5)


def test_gradient_mass_and_gaussian_classification_aids():
    """The two test aids of the full-size gradient tests (oracle/ref_raster.c): blend_bwd(mass=True) is the same
    walk with absolute values, so it bounds |gradient| and equals it when nothing cancels (non-negative upstream
    gradients, colours and depths; v_alpha = 0); gaussian_min_margin returns, per Gaussian, the smallest entry
    of a pixel map over the pixels it (nearly) contributes to."""
    import numpy as np
    from oracle.ref_c import RefC
    from tests.util import scene
    N, W, H, deg = 600, 96, 64, 1
    P, cam = scene(N, W, H, deg, 41)
    R = RefC("f64")
    cb = R.cam_block(cam.viewmat.numpy(), cam.fx, cam.fy, cam.cx, cam.cy, bg=(0.0, 0.0, 0.0))
    Pn = {k: v.double().numpy() for k, v in P.items()}
    pc = R.project_fwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H)
    g, ts = R.bin_sort(pc["rect"], pc["tiles_hit"], pc["depth"], W, H)
    bf = R.blend_fwd(pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g, ts, cb, W, H)
    rng = np.random.default_rng(0)
    args = (pc["xy"], pc["conic"], pc["opac"], pc["rgb"], pc["depth"], g, ts, cb, W, H, bf["final_T"], bf["final_idx"])
    v = (rng.standard_normal((H, W, 3)), rng.standard_normal((H, W)), rng.standard_normal((H, W)))
    bb, m = R.blend_bwd(*args, *v), R.blend_bwd(*args, *v, mass=True)
    for k in bb:
        assert (m[k] >= np.abs(bb[k]) * (1 - 1e-12)).all(), k
        assert (m[k] > 0).sum() == (bb[k] != 0).sum() or k in ("v_conic", "v_xy"), k
    # no cancellation in v_rgb / v_depth when the upstream gradients are non-negative: mass == gradient
    vp = (np.abs(v[0]), np.abs(v[1]), np.zeros((H, W)))
    bp, mp = R.blend_bwd(*args, *vp), R.blend_bwd(*args, *vp, mass=True)
    assert np.allclose(mp["v_rgb"], bp["v_rgb"], rtol=1e-12) and np.allclose(mp["v_depth"], bp["v_depth"], rtol=1e-12)
    # classification: a map that is 1 everywhere except one pixel -> only Gaussians reaching that pixel see 0
    pm = np.ones((H, W)); py, px = H // 2, W // 2; pm[py, px] = 0.0
    gmin, npix = R.gaussian_min_margin(pc["xy"], pc["conic"], pc["opac"], pc["rect"], pc["tiles_hit"], cb, W, H, pm, near=0.5)
    dx, dy = pc["xy"][:, 0] - (px + 0.5), pc["xy"][:, 1] - (py + 0.5)
    sig = 0.5 * (pc["conic"][:, 0] * dx * dx + pc["conic"][:, 2] * dy * dy) + pc["conic"][:, 1] * dx * dy
    tx, ty = px // 16, py // 16
    in_rect = (pc["tiles_hit"] > 0) & (pc["rect"][:, 0] <= tx) & (tx < pc["rect"][:, 2]) & (pc["rect"][:, 1] <= ty) & (ty < pc["rect"][:, 3])
    hits = in_rect & (sig >= 0) & (pc["opac"] * np.exp(-sig) >= 0.5 / 255)
    assert hits.sum() > 0 and np.array_equal(gmin == 0.0, hits)
    assert ((npix > 0) == (gmin < 1e29)).all() and (gmin[(npix > 0) & ~hits] == 1.0).all()
