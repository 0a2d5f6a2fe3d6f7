"""INRIA ``diff_gaussian_rasterization``-shaped surface over the HIP kernels.

``GaussianRasterizationSettings`` / ``GaussianRasterizer`` / ``rasterize_gaussians`` keep the
argument names and meaning of the 3DGS reference rasterizer so that a Touch-GS style training loop
written against it runs unchanged.  Conventions handled here:
* ``viewmatrix`` is the transposed (row-vector) world->view matrix; ``tanfovx/y`` give
  fx = W / (2 tanfovx), cx = W/2 (for a centred principal point the INRIA pixel convention
  ((ndc+1) W - 1)/2 with integer pixel coordinates equals fx x/z + cx with pixel centres at +0.5);
* ``opacities`` and ``scales`` are post-activation, ``rotations`` need not be normalised;
* ``means2D`` is a dummy [N,3] tensor whose ``.grad`` receives the screen-space mean gradient
  (pixel units, columns 0:2) for densification;
* the normative rules of SURVEY App. B (tile rect, 0.3 px blur, 1/255 alpha cut, T <= 1e-4 stop)
  apply;
* ``cov3D_precomp`` ([N,6]: xx, xy, xz, yy, yz, zz; used as given, ``scale_modifier`` is not applied to it -- the INRIA
  rule) is split into scales and a rotation by ``torch.linalg.eigh`` and then takes the same kernels.  The render is the
  one of the equivalent scales + rotations; the gradient w.r.t. the covariance flows back through ``eigh`` and is, like
  ``eigh``'s own, ill-conditioned for Gaussians with (nearly) equal axes.  Touch-GS / Splatfacto never take this path.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from . import ops
from .camera import Camera

SH_C0 = 0.28209479177387814


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def _camera_from_settings(rs: GaussianRasterizationSettings) -> Camera:
    W, H = int(rs.image_width), int(rs.image_height)
    V = rs.viewmatrix.detach().cpu().double().numpy().T  # column-vector world->camera
    bg = tuple(float(b) for b in rs.bg.detach().cpu().tolist())
    return Camera(V, W / (2.0 * rs.tanfovx), H / (2.0 * rs.tanfovy), W / 2.0, H / 2.0, W, H, bg=bg,
                  glob_scale=float(rs.scale_modifier))


def _scales_rotations_from_cov3d(cov6: torch.Tensor):
    """Sigma = R diag(s^2) R^T  ->  (s [N,3], quaternion (w, x, y, z) [N,4]); differentiable (torch.linalg.eigh)."""
    c = cov6
    S = torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1), torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                     torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], -2)
    lam, Q = torch.linalg.eigh(S)
    Q = Q * torch.where(torch.linalg.det(Q) < 0, -1.0, 1.0).to(Q.dtype)[:, None, None]   # proper rotation (det -Q = -det Q in 3-D)
    s = torch.sqrt(torch.clamp(lam, min=1e-20))
    # rotation matrix -> quaternion, the branch with the largest pivot (no division by a small number)
    m00, m11, m22 = Q[:, 0, 0], Q[:, 1, 1], Q[:, 2, 2]
    t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    k = t.argmax(-1)
    r = torch.sqrt(torch.clamp(t.gather(-1, k[:, None])[:, 0], min=1e-20)) * 2      # 4 x the dominant component
    a, b, cc = Q[:, 2, 1] - Q[:, 1, 2], Q[:, 0, 2] - Q[:, 2, 0], Q[:, 1, 0] - Q[:, 0, 1]
    d, e, f = Q[:, 0, 1] + Q[:, 1, 0], Q[:, 0, 2] + Q[:, 2, 0], Q[:, 1, 2] + Q[:, 2, 1]
    cand = torch.stack([torch.stack([r * r / 4, a, b, cc], -1), torch.stack([a, r * r / 4, d, e], -1),
                        torch.stack([b, d, r * r / 4, f], -1), torch.stack([cc, e, f, r * r / 4], -1)], 1)   # x r
    q = cand.gather(1, k[:, None, None].expand(-1, 1, 4))[:, 0] / r[:, None]
    return s, q


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings: GaussianRasterizationSettings, budget: Optional[ops.IntersectBudget] = None):
    """-> (color [3,H,W], radii [N], depth [H,W] expected depth, alpha [H,W])."""
    if (sh is None) == (colors_precomp is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    cam = _camera_from_settings(raster_settings)
    if cov3Ds_precomp is not None and cov3Ds_precomp.numel() > 0:
        if cov3Ds_precomp.shape != (means3D.shape[0], 6):
            raise ValueError("cov3D_precomp must be [N, 6] (xx, xy, xz, yy, yz, zz)")
        scales, rotations = _scales_rotations_from_cov3d(cov3Ds_precomp)
        cam.glob_scale = 1.0       # the INRIA rasterizer applies scale_modifier to scales only, not to a given covariance
    N = means3D.shape[0]
    eps = 1e-7
    o = opacities.reshape(N).clamp(eps, 1 - eps)
    opac_logit = torch.log(o) - torch.log1p(-o)
    log_scales = torch.log(scales)
    if sh is not None:
        deg = int(raster_settings.sh_degree)
        coeffs = sh
    else:  # colours as degree-0 SH: max(C0 * c + 0.5, 0) == colour for colour >= 0
        deg = 0
        coeffs = ((colors_precomp - 0.5) / SH_C0)[:, None, :]
    m2d = means2D[:, :2] if (means2D is not None and means2D.requires_grad) else None
    rgb, depth_acc, alpha, radii = ops.render(means3D, log_scales, rotations, opac_logit, coeffs.contiguous(), cam,
                                              deg, means2d=m2d, budget=budget)
    depth = depth_acc / torch.clamp(alpha, min=1e-10)
    return rgb.permute(2, 0, 1), radii, depth, alpha


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, return_depth: bool = False):
        super().__init__()
        self.raster_settings = raster_settings
        self.return_depth = return_depth

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test used by the INRIA training loop: in front of the near plane."""
        rs = self.raster_settings
        V = rs.viewmatrix.to(positions).T
        z = positions @ V[2, :3] + V[2, 3]
        return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        color, radii, depth, alpha = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                                         rotations, cov3D_precomp, self.raster_settings)
        if self.return_depth:
            return color, radii, depth, alpha
        return color, radii
