"""Camera / frame constants handed to the kernels (C struct TgsCamera)."""
from __future__ import annotations

import dataclasses
from typing import Sequence

import numpy as np


@dataclasses.dataclass
class Camera:
    """World->camera pinhole camera, OpenCV axes (x right, y down, z forward).

    ``viewmat`` is a row-major 4x4 (anything array-like on the host).  nerfstudio cameras are
    camera->world in OpenGL axes: use :meth:`from_c2w_opengl` (same flip as the reference's
    ``utils/create_point_cloud_from_touches.py:64``).
    """
    viewmat: Sequence
    fx: float
    fy: float
    cx: float
    cy: float
    W: int
    H: int
    near: float = 0.01
    pix_center: float = 0.5
    bg: Sequence[float] = (0.0, 0.0, 0.0)
    glob_scale: float = 1.0

    def __post_init__(self):
        v = self.viewmat
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        self.viewmat = np.asarray(v, dtype=np.float64).reshape(4, 4)

    @staticmethod
    def from_c2w_opengl(c2w, fx, fy, cx, cy, W, H, **kw) -> "Camera":
        c2w = np.asarray(c2w.detach().cpu().numpy() if hasattr(c2w, "detach") else c2w, dtype=np.float64)
        m = np.eye(4)
        m[:3, :4] = c2w[:3, :4]
        m[:3, 1:3] *= -1.0  # OpenGL (y up, -z forward) -> OpenCV
        return Camera(np.linalg.inv(m), fx, fy, cx, cy, W, H, **kw)

    def downscaled(self, d: int) -> "Camera":
        """The camera of the image downscaled by the integer factor ``d`` (Splatfacto's resolution schedule;
        nerfstudio ``Cameras.rescale_output_resolution(1 / d)``: focal lengths and principal point / d, sides floored)."""
        if d <= 1:
            return self
        return dataclasses.replace(self, fx=self.fx / d, fy=self.fy / d, cx=self.cx / d, cy=self.cy / d,
                                   W=self.W // d, H=self.H // d)

    @property
    def tiles(self):
        return (self.W + 15) // 16, (self.H + 15) // 16

    @property
    def num_tiles(self):
        tw, th = self.tiles
        return tw * th

    def position(self) -> np.ndarray:
        """Camera centre in world coordinates, -R^T t."""
        R, t = self.viewmat[:3, :3], self.viewmat[:3, 3]
        return -(R.T @ t)

    def c_struct(self):
        from ._lib import TgsCamera
        c = TgsCamera()
        flat = self.viewmat.astype(np.float32).reshape(16)
        for i in range(16):
            c.viewmat[i] = float(flat[i])
        c.fx, c.fy, c.cx, c.cy = float(self.fx), float(self.fy), float(self.cx), float(self.cy)
        c.W, c.H = int(self.W), int(self.H)
        c.near_plane, c.pix_center = float(self.near), float(self.pix_center)
        for i in range(3):
            c.bg[i] = float(self.bg[i])
        c.glob_scale = float(self.glob_scale)
        return c
