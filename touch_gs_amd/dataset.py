"""Scene loading for the self-contained trainer: the on-disk contract the reference's plumbing
produces for ``ns-train depth-gaussian-splatting ... nerfstudio-data``.

``transforms.json``: top-level fl_x, fl_y, cx, cy (, w, h); per frame file_path, transform_matrix
(4x4 camera->world, OpenGL axes), depth_file_path, uncertainty_file_path
(reference utils/transforms_utils.py:40-49, utils/add_depth_file_path_to_transforms.py:37-50).
Depth / uncertainty: 16-bit PNG in millimetres (x 1e-3 -> metres; legacy/dataparser_tactile.py:65-66).
Seed points: points_touch.npy [M,3], points_colors.npy [M,3] in 0..255
(utils/create_point_cloud_from_touches.py:243-244).  Poses are centred and scaled by 1/max|t|
and depths by the same factor (legacy/dataparser_tactile.py:222-235,306,310).

Uncertainty units -- UNVERIFIED-PRIOR.  The fork that loads ``uncertainty_file_path`` is absent from
/root/reference (only the writer is there: utils/add_depth_file_path_to_transforms.py:37-50 registers it
next to ``depth_file_path``, utils/fuse_touch_vision.py:372-376 writes both with the same uint16-mm
encoding).  ``uncertainty_scaling`` chooses what happens to the decoded map:
  "linear"   (default) the same factor as the depth image, depth_unit_scale_factor x dataparser scale --
             what a loader that reuses the depth reader for the second file does (the in-tree
             dataparser has exactly one such reader, legacy/dataparser_tactile.py:159-162,301-312);
  "variance" the square of that factor (the map is a variance of a depth in metres,
             utils/fuse_touch_vision.py:76-202; dimensionally consistent with residual^2 / U);
  "none"     the decoded metres^2 values as stored.
``uncertainty_floor`` (default 0 = off) is a lower bound applied to the decoded map, in the map's own units, before
the scaling: a GPIS touch variance of 0.001 - 0.01 next to the vision prior's >= 5 (utils/fuse_touch_vision.py:310)
gives a touched pixel 500 - 5000 x the weight of any other in the uncertainty-weighted loss -- its residual keeps the
screen-space gradients of the Gaussians it sees above the densification threshold however well they fit (the GP surface
is accurate to millimetres at best, and is rendered into every view), and the refinement piles several 10^5 Gaussians
onto the touched object (56 000 in one tile of a 720p frame: 130 iterations/s instead of 900).  A floor of 0.05 bounds
the ratio at 100.
The choice is recorded by the trainer in config.json (``Scene.describe()``) and used identically by
the nerfstudio plugin's dataparser (nerfstudio_plugin.py).  ``ModelConfig.depth_eps`` is in the
units of the scaled map.

Real-world evaluation (the reference exports IS_REAL_WORLD=True before run_eval,
scripts/train_bunny_real.sh:54, and its aggregator then averages results.gt_depth_mse /
gt_object_depth_mse, experiment_utils/get_results.py:47-51): with ``real_world`` set (default: the
IS_REAL_WORLD environment variable) every view also carries the sensor's depth
``realsense_depths/<n>.png`` (the output of the pipeline's first step) as ground truth and the
touched region ``touch_depth/<n>.png > 0`` as object mask.  The fork that computes the two numbers
is absent; this definition (MSE over measured pixels / over measured object pixels) is the build's.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Tuple

import numpy as np
import torch

from .camera import Camera
from .model import View
from .plumbing import from_uint16_mm, get_train_eval_split_fraction, read_png16


def _read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0


UNCERTAINTY_SCALINGS = ("linear", "variance", "none")


def uncertainty_factor(scaling: str, depth_unit_scale_factor: float, dataparser_scale: float) -> float:
    """Factor applied to the decoded (uint16 mm -> x 1e-3) uncertainty map; see the module docstring."""
    f = depth_unit_scale_factor * 1e3 * dataparser_scale
    return {"linear": f, "variance": f * f, "none": 1.0}[scaling]


class Scene:
    def __init__(self, root: str, train_split_fraction: float = 0.9, device="cuda", scale_poses: bool = True,
                 depth_unit_scale_factor: float = 1e-3, real_world: Optional[bool] = None,
                 gt_depth_dir: str = "realsense_depths", object_mask_dir: str = "touch_depth",
                 uncertainty_scaling: str = "linear", uncertainty_floor: float = 0.0):
        self.root = root
        if uncertainty_scaling not in UNCERTAINTY_SCALINGS:
            raise ValueError(f"uncertainty_scaling must be one of {UNCERTAINTY_SCALINGS}")
        self.uncertainty_scaling = uncertainty_scaling
        self.uncertainty_floor = float(uncertainty_floor)
        self.depth_unit_scale_factor = depth_unit_scale_factor
        if real_world is None:
            real_world = os.environ.get("IS_REAL_WORLD", "").lower() in ("1", "true", "yes")
        self.real_world = real_world
        with open(os.path.join(root, "transforms.json")) as f:
            meta = json.load(f)
        frames = sorted(meta["frames"], key=lambda fr: fr["file_path"])
        c2ws = np.stack([np.array(fr["transform_matrix"], dtype=np.float64) for fr in frames])
        self.scale = 1.0
        if scale_poses:
            c2ws[:, :3, 3] -= c2ws[:, :3, 3].mean(axis=0)
            self.scale = 1.0 / float(np.max(np.abs(c2ws[:, :3, 3])))
            c2ws[:, :3, 3] *= self.scale
        self.offset = None
        self.views: List[View] = []
        for fr, c2w in zip(frames, c2ws):
            rgb = _read_rgb(os.path.join(root, fr["file_path"]))
            H, W = rgb.shape[:2]
            g = lambda k, d=None: fr.get(k, meta.get(k, d))
            cam = Camera.from_c2w_opengl(c2w, g("fl_x"), g("fl_y"), g("cx", W / 2), g("cy", H / 2), W, H)
            depth = unc = None
            if "depth_file_path" in fr:
                depth = torch.from_numpy(from_uint16_mm(read_png16(os.path.join(root, fr["depth_file_path"])))
                                         .astype(np.float32) * (depth_unit_scale_factor * 1e3) * self.scale)
            if "uncertainty_file_path" in fr:
                unc_m = from_uint16_mm(read_png16(os.path.join(root, fr["uncertainty_file_path"]))).astype(np.float32)
                if self.uncertainty_floor > 0:     # in the map's own units, before any scaling (module docstring)
                    unc_m = np.maximum(unc_m, np.float32(self.uncertainty_floor))
                unc = torch.from_numpy(unc_m * np.float32(uncertainty_factor(uncertainty_scaling, depth_unit_scale_factor,
                                                                             self.scale)))
            view = View(cam=cam, rgb=torch.from_numpy(rgb).to(device).contiguous(),
                        depth=None if depth is None else depth.to(device).contiguous(),
                        uncertainty=None if unc is None else unc.to(device).contiguous())
            if real_world:
                stem = os.path.splitext(os.path.basename(fr["file_path"]))[0]
                gt_path = os.path.join(root, gt_depth_dir, stem + ".png")
                if os.path.exists(gt_path):
                    from .prepare import resize_bilinear
                    gt = resize_bilinear(from_uint16_mm(read_png16(gt_path)), H, W).astype(np.float32)
                    view.gt_depth = (torch.from_numpy(gt) * (depth_unit_scale_factor * 1e3) * self.scale).to(device)
                    m_path = os.path.join(root, object_mask_dir, stem + ".png")
                    if os.path.exists(m_path):
                        m = read_png16(m_path) > 0
                        if m.shape == (H, W):
                            view.object_mask = torch.from_numpy(m).to(device)
            self.views.append(view)
        names = [fr["file_path"] for fr in frames]
        self.names = names
        self.i_train, self.i_eval = get_train_eval_split_fraction(names, train_split_fraction)
        self._centre = np.stack([np.array(fr["transform_matrix"], dtype=np.float64)[:3, 3] for fr in frames]).mean(0) \
            if scale_poses else np.zeros(3)

    def describe(self) -> dict:
        """What the trainer records in config.json about the units of the supervision maps."""
        return dict(dataparser_scale=self.scale, depth_unit_scale_factor=self.depth_unit_scale_factor,
                    uncertainty_scaling=self.uncertainty_scaling, uncertainty_floor=self.uncertainty_floor,
                    uncertainty_factor=uncertainty_factor(self.uncertainty_scaling, self.depth_unit_scale_factor,
                                                          self.scale))

    def seed_points(self) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        """points_touch.npy / points_colors.npy, moved into the scaled scene frame."""
        p = os.path.join(self.root, "points_touch.npy")
        c = os.path.join(self.root, "points_colors.npy")
        if not (os.path.exists(p) and os.path.exists(c)):
            return None
        pts = (np.load(p).astype(np.float64) - self._centre) * self.scale
        return torch.from_numpy(pts.astype(np.float32)), torch.from_numpy(np.load(c).astype(np.float32))
