"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/ref_raster.c (see its header).

``RefC("f64")`` / ``RefC("f32")`` expose the scalar C restatement with numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False):
    so = [os.path.join(_HERE, f"libref_raster_{p}.so") for p in ("f64", "f32")]
    src = os.path.join(_HERE, "ref_raster.c")
    if force or not all(os.path.exists(s) and os.path.getmtime(s) >= os.path.getmtime(src) for s in so):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return so


class RefC:
    def __init__(self, prec: str = "f64"):
        path = os.path.join(_HERE, f"libref_raster_{prec}.so")
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        self.dt = np.float64 if prec == "f64" else np.float32
        assert self.lib.ref_real_bytes() == np.dtype(self.dt).itemsize

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def _r(self, a):
        return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=self.dt))

    def cam_block(self, viewmat, fx, fy, cx, cy, near=0.01, pix_center=0.5, bg=(0, 0, 0)):
        v = np.asarray(viewmat, dtype=np.float64).reshape(16)
        return np.ascontiguousarray(np.concatenate([v, [fx, fy, cx, cy, near, pix_center], bg]).astype(self.dt))

    def project_fwd(self, means, log_scales, quats, opac_logit, sh, sh_deg, cam, W, H, glob_scale=1.0):
        means, log_scales, quats, opac_logit, sh = map(self._r, (means, log_scales, quats, opac_logit, sh))
        N = means.shape[0]
        Ks = sh.shape[1] if sh is not None else 0
        o = dict(xy=np.zeros((N, 2), self.dt), depth=np.zeros(N, self.dt), radius=np.zeros(N, np.int32),
                 conic=np.zeros((N, 3), self.dt), rgb=np.zeros((N, 3), self.dt), opac=np.zeros(N, self.dt),
                 rect=np.zeros((N, 4), np.int32), tiles_hit=np.zeros(N, np.int32))
        gs = C.c_double(glob_scale) if self.dt == np.float64 else C.c_float(glob_scale)
        self.lib.ref_project_fwd(C.c_int(N), self._p(means), self._p(log_scales), self._p(quats),
                                 self._p(opac_logit), self._p(sh), C.c_int(Ks), C.c_int(sh_deg),
                                 self._p(cam), C.c_int(W), C.c_int(H), gs,
                                 *[self._p(o[k]) for k in ("xy", "depth", "radius", "conic", "rgb", "opac", "rect", "tiles_hit")])
        return o

    def bin_sort(self, rect, tiles_hit, depth, W, H):
        TW, TH = (W + 15) // 16, (H + 15) // 16
        rect = np.ascontiguousarray(rect, np.int32)
        tiles_hit = np.ascontiguousarray(tiles_hit, np.int32)
        d32 = np.ascontiguousarray(depth, np.float32)
        I = int(tiles_hit.astype(np.int64).sum())
        gid = np.zeros(max(I, 1), np.int32)
        ts = np.zeros(TW * TH + 1, np.int64)
        self.lib.ref_bin_sort(C.c_int(rect.shape[0]), self._p(rect), self._p(tiles_hit), self._p(d32),
                              C.c_int(TW), C.c_int(TH), self._p(gid), self._p(ts))
        return gid[:I], ts

    def num_threads(self):
        return int(self.lib.ref_num_threads())

    def blend_fwd(self, xy, conic, opac, rgb, depth, gid, ts, cam, W, H, tile_range=None):
        xy, conic, opac, rgb, depth = map(self._r, (xy, conic, opac, rgb, depth))
        gid = np.ascontiguousarray(gid, np.int32)
        ts = np.ascontiguousarray(ts, np.int64)
        o = dict(rgb=np.zeros((H, W, 3), self.dt), depth_acc=np.zeros((H, W), self.dt),
                 final_T=np.zeros((H, W), self.dt), final_idx=np.zeros((H, W), np.int32))
        t0, t1 = tile_range if tile_range is not None else (0, ((W + 15) // 16) * ((H + 15) // 16))
        self.lib.ref_blend_fwd_range(self._p(xy), self._p(conic), self._p(opac), self._p(rgb), self._p(depth),
                                     self._p(gid), self._p(ts), self._p(cam), C.c_int(W), C.c_int(H),
                                     self._p(o["rgb"]), self._p(o["depth_acc"]), self._p(o["final_T"]),
                                     self._p(o["final_idx"]), C.c_int(t0), C.c_int(t1))
        o["alpha"] = 1 - o["final_T"]
        return o

    def blend_margin(self, xy, conic, opac, gid, ts, cam, W, H):
        """Per-pixel decision margin (see ref_blend_margin_range) -> float64 [H,W]."""
        xy, conic, opac = map(self._r, (xy, conic, opac))
        gid = np.ascontiguousarray(gid, np.int32)
        ts = np.ascontiguousarray(ts, np.int64)
        m = np.zeros((H, W), np.float64)
        self.lib.ref_blend_margin_range(self._p(xy), self._p(conic), self._p(opac), self._p(gid), self._p(ts),
                                        self._p(cam), C.c_int(W), C.c_int(H), self._p(m), C.c_int(0),
                                        C.c_int(((W + 15) // 16) * ((H + 15) // 16)))
        return m

    def dropped_pairs_max_alpha(self, xy, conic, opac, rect, tiles_hit, tight, cam, W, H):
        """-> (max o*exp(-sigma) per Gaussian over the tiles of `rect` outside `tight`, #dropped pairs)."""
        xy, conic, opac = map(self._r, (xy, conic, opac))
        rect = np.ascontiguousarray(rect, np.int32)
        tight = np.ascontiguousarray(tight, np.int32)
        tiles_hit = np.ascontiguousarray(tiles_hit, np.int32)
        out = np.zeros(xy.shape[0], self.dt)
        cnt = np.zeros(1, np.int64)
        self.lib.ref_dropped_pairs_max_alpha(C.c_int(xy.shape[0]), self._p(xy), self._p(conic), self._p(opac),
                                             self._p(rect), self._p(tiles_hit), self._p(tight), self._p(cam),
                                             C.c_int(W), C.c_int(H), self._p(out), self._p(cnt))
        return out, int(cnt[0])

    def gaussian_min_margin(self, xy, conic, opac, rect, tiles_hit, cam, W, H, pix_margin, near=0.5):
        """-> (min of pix_margin over the pixels each Gaussian (nearly) contributes to [N] float64, their number [N])."""
        xy, conic, opac = map(self._r, (xy, conic, opac))
        rect = np.ascontiguousarray(rect, np.int32)
        tiles_hit = np.ascontiguousarray(tiles_hit, np.int32)
        pm = np.ascontiguousarray(pix_margin, np.float64)
        out = np.zeros(xy.shape[0], np.float64)
        npix = np.zeros(xy.shape[0], np.int64)
        self.lib.ref_gaussian_min_margin(C.c_int(xy.shape[0]), self._p(xy), self._p(conic), self._p(opac),
                                         self._p(rect), self._p(tiles_hit), self._p(cam), C.c_int(W), C.c_int(H),
                                         self._p(pm), C.c_double(near), self._p(out), self._p(npix))
        return out, npix

    def pairs_max_alpha(self, gid, tile, xy, conic, opac, cam, W, H):
        """-> max o*exp(-sigma) over the pixel centres of tile[k] for Gaussian gid[k], per pair."""
        xy, conic, opac = map(self._r, (xy, conic, opac))
        gid = np.ascontiguousarray(gid, np.int32)
        tile = np.ascontiguousarray(tile, np.int32)
        out = np.zeros(len(gid), self.dt)
        self.lib.ref_pairs_max_alpha(C.c_int64(len(gid)), self._p(gid), self._p(tile), self._p(xy), self._p(conic),
                                     self._p(opac), self._p(cam), C.c_int(W), C.c_int(H), self._p(out))
        return out

    def blend_bwd(self, xy, conic, opac, rgb, depth, gid, ts, cam, W, H, final_T, final_idx,
                  v_rgb_img, v_depth_img, v_alpha_img, tile_range=None, mass=False):
        """B.7 -> per-Gaussian screen-space gradients; ``mass=True``: the un-cancelled magnitude of each of them
        (ref_blend_bwd_mass_range: every term entered with its absolute value)."""
        xy, conic, opac, rgb, depth, final_T, v_rgb_img, v_depth_img, v_alpha_img = map(
            self._r, (xy, conic, opac, rgb, depth, final_T, v_rgb_img, v_depth_img, v_alpha_img))
        gid = np.ascontiguousarray(gid, np.int32)
        ts = np.ascontiguousarray(ts, np.int64)
        final_idx = np.ascontiguousarray(final_idx, np.int32)
        N = xy.shape[0]
        o = dict(v_xy=np.zeros((N, 2), self.dt), v_conic=np.zeros((N, 3), self.dt), v_opac=np.zeros(N, self.dt),
                 v_rgb=np.zeros((N, 3), self.dt), v_depth=np.zeros(N, self.dt))
        t0, t1 = tile_range if tile_range is not None else (0, ((W + 15) // 16) * ((H + 15) // 16))
        fn = self.lib.ref_blend_bwd_mass_range if mass else self.lib.ref_blend_bwd_range
        fn(self._p(xy), self._p(conic), self._p(opac), self._p(rgb), self._p(depth),
                                     self._p(gid), self._p(ts), self._p(cam), C.c_int(W), C.c_int(H),
                                     self._p(final_T), self._p(final_idx),
                                     self._p(v_rgb_img), self._p(v_depth_img), self._p(v_alpha_img),
                                     *[self._p(o[k]) for k in ("v_xy", "v_conic", "v_opac", "v_rgb", "v_depth")],
                                     C.c_int(t0), C.c_int(t1))
        return o

    def project_bwd(self, means, log_scales, quats, opac_logit, sh, sh_deg, cam, W, H, radius,
                    v_xy, v_conic, v_opac, v_rgb, v_depth, glob_scale=1.0):
        means, log_scales, quats, opac_logit, sh, v_xy, v_conic, v_opac, v_rgb, v_depth = map(
            self._r, (means, log_scales, quats, opac_logit, sh, v_xy, v_conic, v_opac, v_rgb, v_depth))
        radius = np.ascontiguousarray(radius, np.int32)
        N = means.shape[0]
        Ks = sh.shape[1] if sh is not None else 0
        o = dict(v_means=np.zeros((N, 3), self.dt), v_log_scales=np.zeros((N, 3), self.dt),
                 v_quats=np.zeros((N, 4), self.dt), v_opac_logit=np.zeros(N, self.dt),
                 v_sh=np.zeros((N, max(Ks, 1), 3), self.dt))
        gs = C.c_double(glob_scale) if self.dt == np.float64 else C.c_float(glob_scale)
        self.lib.ref_project_bwd(C.c_int(N), self._p(means), self._p(log_scales), self._p(quats),
                                 self._p(opac_logit), self._p(sh), C.c_int(Ks), C.c_int(sh_deg),
                                 self._p(cam), C.c_int(W), C.c_int(H), gs, self._p(radius),
                                 self._p(v_xy), self._p(v_conic), self._p(v_opac), self._p(v_rgb), self._p(v_depth),
                                 *[self._p(o[k]) for k in ("v_means", "v_log_scales", "v_quats", "v_opac_logit", "v_sh")])
        return o
