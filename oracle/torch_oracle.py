"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for the rasterizer.

Vectorised, differentiable (torch.autograd), fp64-capable CPU restatement of the published
3D-Gaussian-Splatting algorithm exactly as fixed in SURVEY.md Appendix B (sections B.0-B.7).
No reference source exists for these rows (reference ``.gitmodules:7-9`` -> empty submodule;
call site ``scripts/train_bunny_real.sh:52``), so every function cites the Appendix-B
paragraph it restates instead of a reference file:line.

Never imported by the product package ``touch_gs_amd``.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional

import numpy as np
import torch

BLOCK = 16  # B.0 tile size
ALPHA_MIN = 1.0 / 255.0  # B.6
ALPHA_MAX = 0.999  # B.6
T_STOP = 1e-4  # B.6
BLUR = 0.3  # B.3
NEAR = 0.01  # B.0

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


@dataclasses.dataclass
class Camera:
    """B.0 camera: V world->camera (x right, y down, z forward), row-major 4x4."""
    viewmat: torch.Tensor  # [4,4]
    fx: float
    fy: float
    cx: float
    cy: float
    W: int
    H: int
    near: float = NEAR
    pix_center: float = 0.5
    bg: tuple = (0.0, 0.0, 0.0)

    @property
    def tiles(self):
        return (self.W + BLOCK - 1) // BLOCK, (self.H + BLOCK - 1) // BLOCK

    def campos(self, dtype=torch.float64):
        V = self.viewmat.to(dtype)
        return -(V[:3, :3].T @ V[:3, 3])


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """B.5 real SH basis, [N,3] unit dirs -> [N,(deg+1)^2]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    out = [torch.full_like(x, SH_C0)]
    if deg >= 1:
        out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        out += [SH_C2[0] * x * y, SH_C2[1] * y * z, SH_C2[2] * (2 * zz - xx - yy),
                SH_C2[3] * x * z, SH_C2[4] * (xx - yy)]
    if deg >= 3:
        out += [SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * x * y * z,
                SH_C3[2] * y * (4 * zz - xx - yy), SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
                SH_C3[6] * x * (xx - 3 * yy)]
    return torch.stack(out, dim=1)


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """B.2: q=(w,x,y,z) un-normalised -> R(q/|q|), [N,3,3]."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def project(means, log_scales, quats, opac_logit, sh, cam: Camera, sh_deg: int,
            glob_scale: float = 1.0):
    """B.1-B.5.  Returns a dict of per-Gaussian tensors; culled Gaussians have radius 0.

    ``sh`` is [N,K,3] with K >= (sh_deg+1)^2 (only the first (sh_deg+1)^2 are used), or None.
    """
    dt = means.dtype
    N = means.shape[0]
    V = cam.viewmat.to(dt)
    Rw, tw = V[:3, :3], V[:3, 3]
    t = means @ Rw.T + tw  # B.1
    tz = t[:, 2]
    front = tz > cam.near
    tz_s = torch.where(front, tz, torch.ones_like(tz))  # keep culled rows finite
    tx, ty = t[:, 0], t[:, 1]

    Rq = quat_to_rot(quats)  # B.2
    s = torch.exp(log_scales) * glob_scale
    M = Rq * s[:, None, :]
    Sigma = M @ M.transpose(1, 2)

    lim_x = 1.3 * (cam.W / 2) / cam.fx  # B.3
    lim_y = 1.3 * (cam.H / 2) / cam.fy
    txc = tz_s * torch.clamp(tx / tz_s, -lim_x, lim_x)
    tyc = tz_s * torch.clamp(ty / tz_s, -lim_y, lim_y)
    zero = torch.zeros_like(tz_s)
    J = torch.stack([cam.fx / tz_s, zero, -cam.fx * txc / (tz_s * tz_s),
                     zero, cam.fy / tz_s, -cam.fy * tyc / (tz_s * tz_s)], dim=1).view(-1, 2, 3)
    Tm = J @ Rw
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    c00 = cov[:, 0, 0] + BLUR
    c01 = cov[:, 0, 1]
    c11 = cov[:, 1, 1] + BLUR
    det = c00 * c11 - c01 * c01
    ok = front & (det > 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c11 / det_s, -c01 / det_s, c00 / det_s], dim=1)
    mid = 0.5 * (c00 + c11)
    lam1 = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam1.detach())).to(torch.int64)
    xy = torch.stack([cam.fx * tx / tz_s + cam.cx, cam.fy * ty / tz_s + cam.cy], dim=1)

    # B.4 tile rect: truncation toward zero after the divide, then clamp
    TW, TH = cam.tiles
    u, v = xy[:, 0].detach(), xy[:, 1].detach()
    rf = radius.to(dt)
    x0 = torch.clamp(torch.trunc((u - rf) / BLOCK).to(torch.int64), 0, TW)
    x1 = torch.clamp(torch.trunc((u + rf) / BLOCK).to(torch.int64) + 1, 0, TW)
    y0 = torch.clamp(torch.trunc((v - rf) / BLOCK).to(torch.int64), 0, TH)
    y1 = torch.clamp(torch.trunc((v + rf) / BLOCK).to(torch.int64) + 1, 0, TH)
    tiles_hit = (x1 - x0) * (y1 - y0)
    ok = ok & (tiles_hit > 0)
    tiles_hit = torch.where(ok, tiles_hit, torch.zeros_like(tiles_hit))
    radius = torch.where(ok, radius, torch.zeros_like(radius))

    out = dict(xy=xy, depth=tz, conic=conic, radius=radius, tiles_hit=tiles_hit, valid=ok,
               rect=torch.stack([x0, y0, x1, y1], dim=1),
               opac=torch.sigmoid(opac_logit))
    if sh is not None:  # B.5
        campos = cam.campos(dt)
        d = means - campos
        d = d / d.norm(dim=1, keepdim=True)
        K = (sh_deg + 1) ** 2
        Y = sh_basis(sh_deg, d)
        rgb = torch.clamp((Y[:, :, None] * sh[:, :K, :]).sum(1) + 0.5, min=0.0)
        out["rgb"] = rgb
    return out


def depth_sort_key(depth: torch.Tensor) -> np.ndarray:
    """B.6: the IEEE-754 bits of a positive fp32 depth are order preserving."""
    return depth.detach().to(torch.float32).numpy().view(np.uint32).astype(np.uint64)


def bin_and_sort(rect: torch.Tensor, valid: torch.Tensor, depth: torch.Tensor, cam: Camera,
                 exact_depth: bool = False):
    """B.6 keys + stable sort.  Returns (sorted_gid int64[I], tile_start int64[T+1]).

    Ties on depth break by Gaussian index.  ``exact_depth`` sorts on the tensor's own dtype
    instead of fp32 bits (for fp64 end-to-end runs).
    """
    TW, TH = cam.tiles
    rect = rect.numpy()
    gids = np.nonzero(valid.numpy())[0]
    tl, gl = [], []
    for g in gids:
        x0, y0, x1, y1 = rect[g]
        ty, tx = np.meshgrid(np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
        tid = (ty * TW + tx).ravel()
        tl.append(tid)
        gl.append(np.full(tid.shape, g, dtype=np.int64))
    if tl:
        tile = np.concatenate(tl)
        gid = np.concatenate(gl)
    else:
        tile = np.zeros(0, np.int64)
        gid = np.zeros(0, np.int64)
    dkey = depth.detach().numpy()[gid] if exact_depth else depth_sort_key(depth)[gid]
    order = np.lexsort((gid, dkey, tile))
    tile, gid = tile[order], gid[order]
    T = TW * TH
    tile_start = np.searchsorted(tile, np.arange(T + 1), side="left").astype(np.int64)
    return gid, tile_start


def blend(xy, conic, opac, rgb, depth, sorted_gid: np.ndarray, tile_start: np.ndarray,
          cam: Camera, want_margin: bool = False):
    """B.6 front-to-back compositing of RGB and depth in one pass.

    Returns dict(rgb [H,W,3] incl. background, depth_acc [H,W], alpha [H,W], final_T [H,W],
    final_idx int64 [H,W] = position (within the tile list, 0-based) of the last contributor,
    -1 if none).  Differentiable w.r.t. xy, conic, opac, rgb, depth.
    ``margin`` (optional) = per-pixel min relative distance of any threshold test from its
    threshold; pixels with a tiny margin are decision-ambiguous under fp32 rounding.
    """
    dt = xy.dtype
    W, H = cam.W, cam.H
    TW, TH = cam.tiles
    bg = torch.tensor(cam.bg, dtype=dt)
    out_rgb = torch.zeros(H, W, 3, dtype=dt) + bg
    out_d = torch.zeros(H, W, dtype=dt)
    out_T = torch.ones(H, W, dtype=dt)
    out_idx = torch.full((H, W), -1, dtype=torch.int64)
    out_margin = torch.full((H, W), float("inf"), dtype=torch.float64)
    rgb_parts, d_parts, T_parts = {}, {}, {}
    for ty in range(TH):
        for tx in range(TW):
            t = ty * TW + tx
            s, e = int(tile_start[t]), int(tile_start[t + 1])
            if e <= s:
                continue
            g = torch.from_numpy(sorted_gid[s:e])
            ys = torch.arange(ty * BLOCK, min((ty + 1) * BLOCK, H))
            xs = torch.arange(tx * BLOCK, min((tx + 1) * BLOCK, W))
            py, px = torch.meshgrid(ys, xs, indexing="ij")
            pxf = (px.reshape(-1).to(dt) + cam.pix_center)  # [P]
            pyf = (py.reshape(-1).to(dt) + cam.pix_center)
            dx = xy[g, 0][:, None] - pxf[None, :]  # [n,P]  Delta = mean2d - p
            dy = xy[g, 1][:, None] - pyf[None, :]
            a, b, c = conic[g, 0][:, None], conic[g, 1][:, None], conic[g, 2][:, None]
            sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
            araw = opac[g][:, None] * torch.exp(-sigma)
            # B.6 value min(0.999, o e^-sigma); B.7 gradient: "no special-casing of the 0.999 clamp",
            # i.e. the clamp is straight-through (v_sigma = -o e^-sigma v_alpha, v_o = e^-sigma
            # v_alpha also where the clamp is active) -- torch.clamp alone would zero it there.
            alpha = araw + (torch.clamp(araw, max=ALPHA_MAX) - araw).detach()
            skip = (sigma.detach() < 0) | (alpha.detach() < ALPHA_MIN)
            a_eff = torch.where(skip, torch.zeros_like(alpha), alpha)
            Tp = torch.cumprod(1 - a_eff, dim=0)  # T' after each element
            Tb = torch.cat([torch.ones_like(Tp[:1]), Tp[:-1]], dim=0)  # T before
            live = (Tp.detach() > T_STOP)  # monotone: once stopped, stays stopped
            inc = live & ~skip
            wgt = torch.where(inc, a_eff * Tb, torch.zeros_like(a_eff))
            Cpix = wgt.T @ rgb[g]  # [P,3]
            Dpix = wgt.T @ depth[g]
            # final T = product over included elements
            Tfin = torch.prod(torch.where(inc, 1 - a_eff, torch.ones_like(a_eff)), dim=0)
            n = e - s
            pos = torch.arange(n)[:, None].expand(n, pxf.shape[0])
            last = torch.where(inc, pos, torch.full_like(pos, -1)).max(dim=0).values
            hh, ww = ys.numel(), xs.numel()
            sl = (slice(ty * BLOCK, ty * BLOCK + hh), slice(tx * BLOCK, tx * BLOCK + ww))
            rgb_parts[t] = (sl, Cpix.view(hh, ww, 3) + Tfin.view(hh, ww, 1) * bg)
            d_parts[t] = Dpix.view(hh, ww)
            T_parts[t] = Tfin.view(hh, ww)
            out_idx[sl] = last.view(hh, ww)
            if want_margin:
                # only tests actually evaluated (i.e. before the stop) matter
                evald = torch.cat([torch.ones_like(live[:1]), live[:-1]], dim=0)
                m_a = ((araw.detach() - ALPHA_MIN).abs() / ALPHA_MIN).double()
                m_t = ((Tp.detach() - T_STOP).abs() / T_STOP).double()
                m_t = torch.where(skip, torch.full_like(m_t, float("inf")), m_t)
                m = torch.minimum(m_a, m_t)
                m = torch.where(evald, m, torch.full_like(m, float("inf")))
                out_margin[sl] = m.min(dim=0).values.view(hh, ww)
    # assemble differentiably
    rgb_img, d_img, T_img = out_rgb, out_d, out_T
    if rgb_parts:
        rgb_img = out_rgb.clone()
        d_img = out_d.clone()
        T_img = out_T.clone()
        for t, (sl, vv) in rgb_parts.items():
            rgb_img[sl] = vv
            d_img[sl] = d_parts[t]
            T_img[sl] = T_parts[t]
    res = dict(rgb=rgb_img, depth_acc=d_img, alpha=1 - T_img, final_T=T_img, final_idx=out_idx)
    if want_margin:
        res["margin"] = out_margin
    return res


def render(means, log_scales, quats, opac_logit, sh, cam: Camera, sh_deg: int,
           glob_scale: float = 1.0, want_margin: bool = False, exact_depth: bool = False):
    """Full B.1-B.6 pipeline.  Returns (blend dict, projection dict, sorted_gid, tile_start)."""
    pr = project(means, log_scales, quats, opac_logit, sh, cam, sh_deg, glob_scale)
    gid, tstart = bin_and_sort(pr["rect"], pr["valid"], pr["depth"], cam, exact_depth)
    out = blend(pr["xy"], pr["conic"], pr["opac"], pr["rgb"], pr["depth"], gid, tstart, cam,
                want_margin)
    return out, pr, gid, tstart


# ---------------------------------------------------------------------------------------
# a11 tactile depth / uncertainty loss (build-defined; flag names from the reference's
# scripts/train_bunny_real.sh:52, train_block_data.sh:50, train_bunny_blender.sh:50).
# ---------------------------------------------------------------------------------------

def depth_loss(depth_acc, alpha, d_gt, unc, loss_type: str, uncertainty_weight: float,
               eps: float = 1e-6, alpha_eps: float = 1e-10):
    """SURVEY section 8 row a11.  D_hat = depth_acc / max(alpha, alpha_eps); valid m = D_gt > 0.

    SIMPLE_LOSS:                      mean_m (D_hat - D_gt)^2
    DEPTH_UNCERTAINTY_WEIGHTED_LOSS:  mean_m (D_hat - D_gt)^2 / (uncertainty_weight * U + eps)
    (mean over valid pixels; 0 if there are none).
    """
    m = d_gt > 0
    cnt = m.sum()
    if cnt == 0:
        return depth_acc.sum() * 0
    dhat = depth_acc / torch.clamp(alpha, min=alpha_eps)
    r2 = (dhat - d_gt) ** 2
    if loss_type == "DEPTH_UNCERTAINTY_WEIGHTED_LOSS":
        r2 = r2 / (uncertainty_weight * unc + eps)
    elif loss_type != "SIMPLE_LOSS":
        raise ValueError(loss_type)
    return torch.where(m, r2, torch.zeros_like(r2)).sum() / cnt


def l1_loss(rgb, gt):
    return (rgb - gt).abs().mean()


def gaussian_window(size: int = 11, sigma: float = 1.5, dtype=torch.float64):
    x = torch.arange(size, dtype=dtype) - size // 2
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    return g / g.sum()


def ssim(img1, img2, size: int = 11, sigma: float = 1.5):
    """Mean SSIM of two [H,W,3] images in [0,1]; 11x11 Gaussian window sigma 1.5, zero padding
    ('same' convolution, as in the 3DGS training code), C1=0.01^2, C2=0.03^2."""
    import torch.nn.functional as F
    g = gaussian_window(size, sigma, img1.dtype)
    w2 = (g[:, None] * g[None, :])[None, None].expand(3, 1, size, size)
    a = img1.permute(2, 0, 1)[None]
    b = img2.permute(2, 0, 1)[None]
    pad = size // 2
    mu1 = F.conv2d(a, w2, padding=pad, groups=3)
    mu2 = F.conv2d(b, w2, padding=pad, groups=3)
    s11 = F.conv2d(a * a, w2, padding=pad, groups=3) - mu1 * mu1
    s22 = F.conv2d(b * b, w2, padding=pad, groups=3) - mu2 * mu2
    s12 = F.conv2d(a * b, w2, padding=pad, groups=3) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def train_loss(out, gt_rgb, d_gt, unc, ssim_lambda=0.2, depth_loss_mult=0.0,
               depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS", uncertainty_weight=1.0):
    """a10/a11: (1-l)*L1 + l*(1-SSIM) + depth_loss_mult*L_depth."""
    L = (1 - ssim_lambda) * l1_loss(out["rgb"], gt_rgb)
    if ssim_lambda > 0:
        L = L + ssim_lambda * (1 - ssim(out["rgb"], gt_rgb))
    if depth_loss_mult > 0:
        L = L + depth_loss_mult * depth_loss(out["depth_acc"], out["alpha"], d_gt, unc,
                                             depth_loss_type, uncertainty_weight)
    return L


# ---------------------------------------------------------------------------------------
# Synthetic scene S(N,W,H,deg,seed) of SURVEY section 8d (oracle-side restatement; the product
# has its own generator in touch_gs_amd/scene.py -- tests check they agree).
# ---------------------------------------------------------------------------------------

def synthetic_scene(N, W, H, deg, seed, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    fx = fy = (W / 2) / math.tan(math.radians(30.0))
    t30 = math.tan(math.radians(30.0))
    u = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    z = 2 + 4 * u(N)
    x = (2 * u(N) - 1) * 1.1 * z * t30
    y = (2 * u(N) - 1) * 1.1 * z * (H / W) * t30
    means = torch.stack([x, y, z], 1)
    log_scales = math.log(7.0 / fx) + 0.6 * n(N, 3)
    quats = n(N, 4)
    opac = -2 + 4 * u(N)
    K = (deg + 1) ** 2
    sh = torch.zeros(N, K, 3, dtype=torch.float64)
    sh[:, 0, :] = (u(N, 3) - 0.5) / SH_C0
    if K > 1:
        sh[:, 1:, :] = 0.05 * n(N, K - 1, 3)
    P = dict(means=means.to(dtype), log_scales=log_scales.to(dtype), quats=quats.to(dtype),
             opac_logit=opac.to(dtype), sh=sh.to(dtype))
    return P, dict(fx=fx, fy=fy, cx=W / 2, cy=H / 2, W=W, H=H)


def orbit_viewmat(k: int, V: int, centre=(0.0, 0.0, 4.0), dtype=torch.float64):
    """View k of V: rotation about the y axis through the scene centre by 2*pi*k/V (view 0 = I)."""
    th = 2 * math.pi * k / V
    c, s = math.cos(th), math.sin(th)
    R = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=dtype)
    ctr = torch.tensor(centre, dtype=dtype)
    M = torch.eye(4, dtype=dtype)
    M[:3, :3] = R
    M[:3, 3] = ctr - R @ ctr
    return M
