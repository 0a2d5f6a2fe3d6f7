"""Synthetic scene S(N, W, H, deg, seed) of SURVEY section 8(d) -- the workload bench.py measures.

The reference's scenes live in an absent data submodule (``touch-gs-data``, .gitmodules:1-3), so
throughput is quoted on this seeded generator.  Generation happens on the CPU with a
``torch.Generator`` (bit-reproducible everywhere) and is copied to the device.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

from .camera import Camera

SH_C0 = 0.28209479177387814


def synthetic_gaussians(N: int, W: int, H: int, deg: int, seed: int,
                        clustered: bool = False) -> Tuple[Dict[str, torch.Tensor], dict]:
    """z~U(2,6); x,y fill 1.1x the frustum; log-scales ~ N(log(7/fx), 0.6^2); quats ~ N(0,1)^4;
    opacity logits ~ U(-2,2); SH dc = (U(0,1)-0.5)/C0, higher bands ~ N(0, 0.05^2).

    ``clustered`` (stress scene, not a BASELINE config): 80 % of the Gaussians are squeezed into the
    central 10 % of the image (x, y scaled by sqrt(0.1) of the frustum) -- the object-centric shape
    of the reference's scenes (one object on a table, scripts/train_bunny_real.sh), which puts
    thousands of Gaussians into a few tiles.  Same random draws as the uniform scene."""
    g = torch.Generator().manual_seed(seed)
    t30 = math.tan(math.radians(30.0))
    fx = fy = (W / 2) / t30
    u = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    z = 2 + 4 * u(N)
    x = (2 * u(N) - 1) * 1.1 * z * t30
    y = (2 * u(N) - 1) * 1.1 * z * (H / W) * t30
    if clustered:
        inner = (torch.arange(N) % 5) != 0          # 80 %
        k = math.sqrt(0.1) / 1.1
        x = torch.where(inner, x * k, x)
        y = torch.where(inner, y * k, y)
    means = torch.stack([x, y, z], 1)
    log_scales = math.log(7.0 / fx) + 0.6 * n(N, 3)
    quats = n(N, 4)
    opac = -2 + 4 * u(N)
    K = (deg + 1) ** 2
    sh = torch.zeros(N, K, 3, dtype=torch.float64)
    sh[:, 0, :] = (u(N, 3) - 0.5) / SH_C0
    if K > 1:
        sh[:, 1:, :] = 0.05 * n(N, K - 1, 3)
    P = dict(means=means.float(), log_scales=log_scales.float(), quats=quats.float(),
             opac_logit=opac.float(), sh=sh.float())
    return P, dict(fx=fx, fy=fy, cx=W / 2, cy=H / 2, W=W, H=H)


def orbit_viewmat(k: int, V: int, centre=(0.0, 0.0, 4.0)) -> np.ndarray:
    """View k of V: rotation about the y axis through the scene centre by 2*pi*k/V (view 0 = I)."""
    th = 2 * math.pi * k / V
    c, s = math.cos(th), math.sin(th)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
    ctr = np.asarray(centre, dtype=np.float64)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = ctr - R @ ctr
    return M


def make_camera(intr: dict, view: int = 0, n_views: int = 8, bg=(0.0, 0.0, 0.0)) -> Camera:
    return Camera(orbit_viewmat(view, n_views), intr["fx"], intr["fy"], intr["cx"], intr["cy"],
                  intr["W"], intr["H"], bg=bg)


def supervision_maps(W: int, H: int, seed: int):
    """Uncertainty ~ U(0.001, 5); 30 % of the pixels carry no depth supervision (D_gt = 0)."""
    g = torch.Generator().manual_seed(seed + 777)
    unc = 0.001 + (5 - 0.001) * torch.rand(H, W, generator=g)
    masked = torch.rand(H, W, generator=g) < 0.3
    return unc, masked


def make_view(N: int, W: int, H: int, deg: int, seed: int, device, view: int = 0, n_views: int = 8,
              clustered: bool = False, _scene=None):
    """Ground truth for a view = render of the *other* scene S(.., seed+1000) through this library
    (GT RGB, GT depth), plus the synthetic uncertainty / mask maps."""
    from . import ops
    from .model import View
    P, intr = _scene if _scene is not None else synthetic_gaussians(N, W, H, deg, seed + 1000, clustered=clustered)
    cam = make_camera(intr, view, n_views)
    D = P if _scene is not None else {k: v.to(device).contiguous() for k, v in P.items()}
    with torch.no_grad():
        rgb, depth_acc, alpha, _ = ops.render(D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                              D["sh"], cam, deg)
        depth = depth_acc / torch.clamp(alpha, min=1e-10)
    unc, masked = supervision_maps(W, H, seed)
    depth = torch.where(masked.to(device), torch.zeros_like(depth), depth)
    return View(cam=cam, rgb=rgb.clamp(0, 1).contiguous(), depth=depth.contiguous(),
                uncertainty=unc.to(device).contiguous())


def write_scene_dir(root: str, views, points=None, colors255=None) -> None:
    """Writes views to disk in the format the reference's plumbing leaves behind, i.e. what
    ``touch_gs_amd.train --data`` and the nerfstudio dataparser read: ``transforms.json`` (fl_x .. h, per frame
    ``file_path`` / OpenGL camera->world ``transform_matrix`` / ``depth_file_path`` / ``uncertainty_file_path``,
    reference utils/add_depth_file_path_to_transforms.py:37-50), 8-bit RGB PNGs under images/, 16-bit millimetre
    depth and uncertainty PNGs (utils/fuse_touch_vision.py:372-376) and, if given, the touch seed points
    ``points_touch.npy`` / ``points_colors.npy`` (utils/create_point_cloud_from_touches.py:243-244)."""
    import json
    import os
    import numpy as np
    from PIL import Image
    from . import plumbing
    for d in ("images", "fused_output_dir", "fused_output_dir_uncertainty"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    frames = []
    for i, v in enumerate(views):
        name = f"{i:04d}.png"
        Image.fromarray((v.rgb.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()).save(os.path.join(root, "images", name))
        plumbing.write_png16(os.path.join(root, "fused_output_dir", name),
                             plumbing.to_uint16_mm(v.depth.cpu().numpy().astype(np.float64)))
        plumbing.write_png16(os.path.join(root, "fused_output_dir_uncertainty", name),
                             plumbing.to_uint16_mm(v.uncertainty.cpu().numpy().astype(np.float64)))
        c2w = np.linalg.inv(np.asarray(v.cam.viewmat, dtype=np.float64)) @ np.diag([1.0, -1.0, -1.0, 1.0])
        frames.append({"file_path": f"images/{name}", "transform_matrix": c2w.tolist()})
    cam = views[0].cam
    meta = {"fl_x": cam.fx, "fl_y": cam.fy, "cx": cam.cx, "cy": cam.cy, "w": cam.W, "h": cam.H, "frames": frames}
    plumbing.add_depth_file_paths(meta, "fused_output_dir", "fused_output_dir_uncertainty")
    with open(os.path.join(root, "transforms.json"), "w") as f:
        json.dump(meta, f)
    if points is not None:
        np.save(os.path.join(root, "points_touch.npy"), np.asarray(points, dtype=np.float64))
        np.save(os.path.join(root, "points_colors.npy"), np.asarray(colors255, dtype=np.float64))


def make_views(N: int, W: int, H: int, deg: int, seed: int, device, n_views: int, clustered: bool = False):
    """``[make_view(.., view=v, n_views=n_views) for v in range(n_views)]`` with the ground-truth scene generated and
    uploaded once (the generator runs on the CPU: ~0.7 s per call at 300 k Gaussians).  -> (views, ground-truth
    parameter dict on the device)."""
    P, intr = synthetic_gaussians(N, W, H, deg, seed + 1000, clustered=clustered)
    D = {k: v.to(device).contiguous() for k, v in P.items()}
    return [make_view(N, W, H, deg, seed, device, view=v, n_views=n_views, clustered=clustered, _scene=(D, intr))
            for v in range(n_views)], D
