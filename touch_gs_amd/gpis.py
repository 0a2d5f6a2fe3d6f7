"""Gaussian-Process Implicit Surface (GPIS): touch points -> per-view depth + variance maps.

SURVEY section 8 row a13.  The reference obtains these maps from an un-vendored submodule
(``gpis/`` = armlabstanford/GPIS, reference .gitmodules:4-6; empty in the tree) and only consumes
its outputs ``gpis_depth/Image<N>.npy`` / ``gpis_var/Image<N>.npy`` -- float arrays in metres /
variance with NaN where no surface is seen (reference utils/read_touch_depths.py:41-49).  The
algorithm below is therefore a DOCUMENTED CHOICE (parity unpinned), the textbook GPIS:

* observations: f = 0 at the touched surface points, f = +d at points offset by d along the
  outward normal and f = -d inside (signed-distance targets);
* zero-mean GP with a squared-exponential kernel, exact inference by Cholesky (n <= a few 10^3);
* per view, every pixel ray in the region of interest is marched through the posterior mean;
  the first outside->inside sign change is refined by bisection; depth = camera-space z of the
  hit, variance = GP posterior variance there; pixels without a crossing are NaN.

CPU / NumPy like the rest of the reference's config[0] plumbing.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
from scipy.linalg import cho_factor, cho_solve


def estimate_outward_normals(points: np.ndarray, sensor_positions: np.ndarray, k: int = 12) -> np.ndarray:
    """PCA normal of the k nearest neighbours, oriented towards the sensor that touched the point."""
    from scipy.spatial import cKDTree
    tree = cKDTree(points)
    _, idx = tree.query(points, k=min(k, len(points)))
    nb = points[idx] - points[idx].mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", nb, nb)
    w, v = np.linalg.eigh(cov)
    n = v[:, :, 0]
    flip = np.einsum("ni,ni->n", n, sensor_positions - points) < 0
    n[flip] *= -1
    return n


class GPIS:
    """Exact GP on signed-distance observations of a touched surface."""

    def __init__(self, length_scale: float = 0.05, signal_var: float = 1.0, noise_var: float = 1e-4,
                 offset: float = 0.02):
        self.l, self.sf2, self.sn2, self.offset = length_scale, signal_var, noise_var, offset
        self.X = None

    def _k(self, A: np.ndarray, B: np.ndarray) -> np.ndarray:
        d2 = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T
        return self.sf2 * np.exp(-0.5 * np.maximum(d2, 0.0) / (self.l * self.l))

    def fit(self, surface_points: np.ndarray, outward_normals: np.ndarray, max_points: int = 1500,
            rng: Optional[np.random.Generator] = None) -> "GPIS":
        P = np.asarray(surface_points, dtype=np.float64)
        Nn = np.asarray(outward_normals, dtype=np.float64)
        Nn = Nn / np.linalg.norm(Nn, axis=1, keepdims=True)
        if len(P) > max_points:
            sel = (rng or np.random.default_rng(0)).choice(len(P), max_points, replace=False)
            P, Nn = P[sel], Nn[sel]
        d = self.offset
        # signed-distance targets along the normal: +d outside, -d and -2d inside (the second inside layer makes the
        # negative shell behind a touched patch thick enough for a ray march to find).  An offset point that lies
        # closer to ANOTHER part of the touched surface than its own offset says (a thin part: the inside point of
        # one face comes out of the opposite face) would contradict that part's own observations: dropped.
        self.P, self.N = P, Nn
        cand = [(P + d * Nn, d), (P - d * Nn, -d), (P - 2 * d * Nn, -2 * d)]
        Xs, ys = [P], [np.zeros(len(P))]
        from scipy.spatial import cKDTree
        tree = cKDTree(P)
        for Q, val in cand:
            near, _ = tree.query(Q)
            ok = near >= 0.9 * abs(val)
            Xs.append(Q[ok])
            ys.append(np.full(int(ok.sum()), val))
        self.X = np.concatenate(Xs)
        y = np.concatenate(ys)
        K = self._k(self.X, self.X)
        K[np.diag_indices_from(K)] += self.sn2
        self.chol = cho_factor(K, lower=True)
        self.alpha = cho_solve(self.chol, y)
        # prior mean of "far outside": keeps rays that never come near a touch strictly positive
        self.prior = d
        return self

    def _predict_torch(self, Q: np.ndarray, want_var: bool):
        """Same arithmetic as the NumPy form, in fp64 torch on ``self.device`` (the kernel matrix of a ray-marching
        step is ~1e8 exponentials: minutes per view in single-threaded NumPy, a fraction of a second threaded / on a GPU)."""
        import torch
        dev = torch.device(self.device)
        if getattr(self, "_t", None) is None or self._t_key != str(self.device):
            self._t_key = str(self.device)   # (not tensor.device == dev: "cuda" and "cuda:0" compare unequal)
            T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
            # K^-1 once on the host (the device BLAS's triangular solve wants a workspace per right-hand side block)
            self._t = (T(self.X), T(self.alpha), T(cho_solve(self.chol, np.eye(len(self.X)))))
        X, alpha, Kinv = self._t
        q = torch.as_tensor(np.ascontiguousarray(Q), dtype=torch.float64, device=dev)
        d2 = (q * q).sum(1)[:, None] + (X * X).sum(1)[None, :] - 2.0 * q @ X.T
        Kq = self.sf2 * torch.exp(-0.5 * d2.clamp_min(0.0) / (self.l * self.l))
        support = (Kq.max(dim=1).values / self.sf2).clamp(0.0, 1.0)
        mean = Kq @ alpha + self.prior * (1.0 - support)
        if not want_var:
            return mean.cpu().numpy(), None
        var = self.sf2 - ((Kq @ Kinv) * Kq).sum(1)
        return mean.cpu().numpy(), var.clamp_min(0.0).cpu().numpy()

    def _tree(self):
        if getattr(self, "_kd", None) is None:
            from scipy.spatial import cKDTree
            self._kd = cKDTree(self.P)
        return self._kd

    def predict(self, Q: np.ndarray, want_var: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        if getattr(self, "device", None) is not None:
            return self._predict_torch(Q, want_var)
        Kq = self._k(np.asarray(Q, dtype=np.float64), self.X)
        # blend towards the positive ("outside") prior where the GP has no support -- a zero-mean GP
        # would otherwise report f = 0 (a surface) everywhere far from the touches
        support = np.clip(Kq.max(axis=1) / self.sf2, 0.0, 1.0)
        mean = Kq @ self.alpha + self.prior * (1.0 - support)
        if not want_var:
            return mean, None
        v = cho_solve(self.chol, Kq.T)
        var = self.sf2 - np.einsum("ij,ji->i", Kq, v)
        return mean, np.maximum(var, 0.0)

    def render_depth(self, c2w_opengl: np.ndarray, fx: float, fy: float, cx: float, cy: float, W: int, H: int,
                     near: float = 0.02, far: float = 2.0, n_steps: Optional[int] = None, stride: int = 1,
                     roi_margin_px: int = 24, chunk: int = 20000, max_var: Optional[float] = None,
                     var_floor: float = 0.0):
        """-> (depth [H,W] metres, var [H,W]); NaN where the ray meets no surface.  ``c2w_opengl`` is
        the transforms.json camera (x right, y up, -z forward).

        ``n_steps``: samples of the march between the nearest and the farthest touched point (+- 3 length scales);
        default: one every 0.6 x offset, so that the negative shell behind a touched patch (~2 offsets thick) cannot be
        stepped over.  A crossing is kept only if the touched point that dominates it FACES the camera (its outward
        normal against the ray): a ray that comes from an untouched side and enters the inside shell of a far-side
        patch from behind crosses from the positive prior to negative values as well, but no surface is seen there.
        ``max_var``: crossings whose posterior variance exceeds it are reported as "no surface seen" (NaN): far from
        every touch the posterior mean is the prior and a zero crossing there carries no information -- without the
        cut, the inverse-variance fusion downstream (variance ~ signal_var = 1 against the vision prior's >= 5,
        utils/fuse_touch_vision.py:76-202, :310) would let such a crossing outvote the monocular depth.
        ``var_floor``: lower bound of the reported variance; the pipeline stores the map as uint16 x 1000
        (utils/read_touch_depths.py:52-53), so a variance below 1e-3 would be truncated to 0 = "no touch here"."""
        c2w = np.asarray(c2w_opengl, dtype=np.float64)
        R = c2w[:3, :3] @ np.diag([1.0, -1.0, -1.0])  # OpenCV camera axes in the world
        t = c2w[:3, 3]
        depth = np.full((H, W), np.nan)
        var = np.full((H, W), np.nan)
        # region of interest: pixels near the projection of the observed surface points
        Pc = (self.P - t) @ R
        front = Pc[:, 2] > near
        if not front.any():
            return depth, var
        u = fx * Pc[front, 0] / Pc[front, 2] + cx
        v = fy * Pc[front, 1] / Pc[front, 2] + cy
        u0, u1 = int(max(0, np.floor(u.min()) - roi_margin_px)), int(min(W - 1, np.ceil(u.max()) + roi_margin_px))
        v0, v1 = int(max(0, np.floor(v.min()) - roi_margin_px)), int(min(H - 1, np.ceil(v.max()) + roi_margin_px))
        if u1 < u0 or v1 < v0:
            return depth, var
        zs_lo = max(near, Pc[front, 2].min() - 3 * self.l - self.offset)
        zs_hi = min(far, Pc[front, 2].max() + 3 * self.l + self.offset)
        us, vs = np.meshgrid(np.arange(u0, u1 + 1, stride), np.arange(v0, v1 + 1, stride))
        us, vs = us.ravel(), vs.ravel()
        dirs_c = np.stack([(us + 0.5 - cx) / fx, (vs + 0.5 - cy) / fy, np.ones_like(us, dtype=np.float64)], 1)
        if n_steps is None:
            n_steps = int(np.clip(np.ceil((zs_hi - zs_lo) / (0.6 * self.offset)), 16, 512))
        zgrid = np.linspace(zs_lo, zs_hi, n_steps)
        for s in range(0, len(us), chunk):
            dc = dirs_c[s:s + chunk]
            n = len(dc)
            dw = dc @ R.T
            prev = np.full(n, np.inf)
            z_lo = np.full(n, np.nan)
            z_hi = np.full(n, np.nan)
            done = np.zeros(n, dtype=bool)
            for z in zgrid:
                m, _ = self.predict(t + z * dw, want_var=False)
                hit = (~done) & (prev > 0) & (m <= 0) & np.isfinite(prev)
                z_lo[hit] = z - (zgrid[1] - zgrid[0])
                z_hi[hit] = z
                done |= hit
                prev = m
            idx = np.nonzero(done)[0]
            if len(idx) == 0:
                continue
            lo, hi = z_lo[idx], z_hi[idx]
            for _ in range(12):  # bisection on the posterior mean
                mid = 0.5 * (lo + hi)
                m, _ = self.predict(t + mid[:, None] * dw[idx], want_var=False)
                inside = m <= 0
                hi = np.where(inside, mid, hi)
                lo = np.where(inside, lo, mid)
            zhit = 0.5 * (lo + hi)
            _, vv = self.predict(t + zhit[:, None] * dw[idx], want_var=True)
            # facing test: outward normal of the touched point nearest to the crossing against the ray
            ph = t + zhit[:, None] * dw[idx]
            _, nn = self._tree().query(ph)
            sure = np.einsum("ij,ij->i", self.N[nn], dw[idx] / np.linalg.norm(dw[idx], axis=1, keepdims=True)) < -0.05
            if max_var is not None:
                sure &= vv <= max_var
            idx, zhit, vv = idx[sure], zhit[sure], vv[sure]
            depth[vs[s:s + chunk][idx], us[s:s + chunk][idx]] = zhit
            var[vs[s:s + chunk][idx], us[s:s + chunk][idx]] = np.maximum(vv, var_floor)
        return depth, var
