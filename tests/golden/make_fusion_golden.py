"""Generates tests/golden/fusion_reference.npz by IMPORTING the reference's own plumbing
(/root/reference/utils/*.py) in this container -- these are REFERENCE outputs, not self-oracle.

cv2 / open3d are not installed; they are stubbed in sys.modules exactly as SURVEY.md section 8(c)
describes (cv2 is only used for file I/O, visualisation and for blur/medianBlur on a term that the
pipeline multiplies by proximity_weight = 0.0, reference utils/fuse_touch_vision.py:310).
Only data (inputs + expected outputs) is written; no reference source is copied.
Run from the repo root:  python tests/golden/make_fusion_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/utils"


def _stub_modules():
    cv2 = types.ModuleType("cv2")
    cv2.blur = lambda a, k: a            # multiplied by proximity_weight = 0 at the only call site
    cv2.medianBlur = lambda a, k: a
    cv2.IMREAD_ANYDEPTH = 2
    cv2.INTER_LINEAR = 1
    sys.modules["cv2"] = cv2
    o3d = types.ModuleType("open3d")
    sys.modules["open3d"] = o3d


def main():
    _stub_modules()
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")
    import fuse_touch_vision as ftv                      # noqa: E402
    import create_uncertainty_from_depth as cu           # noqa: E402
    import create_point_cloud_from_touches as cpc        # noqa: E402

    out = {}
    H, W = 40, 64
    f32 = lambda a: a.astype(np.float32).astype(np.float64)   # inputs are float32-representable
    for seed in range(2):
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:H, 0:W]
        true = 1.5 + 0.6 * np.sin(xx / 17.0 + seed) + 0.4 * np.cos(yy / 11.0) + 0.002 * xx
        vision = (true - 0.12) / 1.07 + 0.01 * rng.standard_normal((H, W))      # mis-scaled monocular depth
        vision = f32(np.clip(vision, 0.05, None))
        grounded_dense = f32(true + 0.005 * rng.standard_normal((H, W)))
        np.random.seed(100 + seed)                                              # the reference's sparsifier is unseeded
        grounded = ftv.create_sparse_depth_map(grounded_dense, keep_percentage=0.01)
        touch = np.zeros((H, W))
        tvar = np.zeros((H, W))
        cy, cx = 14 + 7 * seed, 22 + 11 * seed
        patch = (yy - cy) ** 2 + (xx - cx) ** 2 < 10 ** 2
        touch[patch] = (true - 0.03)[patch]
        tvar[patch] = (0.0005 + 0.004 * rng.random((H, W)))[patch]
        touch, tvar = f32(touch), f32(tvar)
        for real in (True, False):
            tag = f"s{seed}/{'real' if real else 'sim'}"
            ds, va, vu = ftv.align_vision_depth(grounded.copy(), touch.copy(), vision.copy(), is_real_world=real)
            fd, fu = ftv.fuse_depth_maps_with_uncertainty(touch.copy(), va.copy(), tvar.copy(), vu.copy())
            fd = np.clip(fd, 0, None)
            fu = np.clip(fu, 0, 10)
            s1 = ftv.compute_scale_and_offset_best(grounded, vision, None, (0, None), (None, None))
            out[f"{tag}/ds_gs"], out[f"{tag}/vision_aligned"], out[f"{tag}/vision_unc"] = ds, va, vu
            out[f"{tag}/fused_depth"], out[f"{tag}/fused_unc"] = fd, fu
            out[f"{tag}/scale_offset_1"] = np.array(s1)
        out[f"s{seed}/grounded_dense"], out[f"s{seed}/grounded"] = grounded_dense, grounded
        out[f"s{seed}/touch"], out[f"s{seed}/touch_var"], out[f"s{seed}/vision"] = touch, tvar, vision
        # the full uncertainty map with the edge / difference terms switched on (cv2-free terms)
        out[f"s{seed}/unc_edges"] = cu.compute_uncertainty_map_with_edges(
            vision, grounded, edge_weight=0.7, distance_uncertainty_weight=0.1, proximity_weight=0.0,
            dilation_size=2, depth_difference_weight=1.3)
    # known-answer check of the constrained fit: sparse = 1.1*dense + 0.02
    dense = np.linspace(0.5, 3.0, 600).reshape(20, 30)
    sparse = np.where(np.arange(600).reshape(20, 30) % 7 == 0, 1.1 * dense + 0.02, 0.0)
    out["fit/dense"], out["fit/sparse"] = dense, sparse
    out["fit/free"] = np.array(ftv.compute_scale_and_offset_best(sparse, dense, None, (0, None), (None, None)))
    out["fit/nonneg"] = np.array(ftv.compute_scale_and_offset_best(sparse, dense, None, (0, None), (0, None)))
    out["fit/unit_scale"] = np.array(ftv.compute_scale_and_offset_best(sparse, dense, None, (1, 1), (None, None)))
    out["fit/clamped"] = np.array(ftv.compute_scale_and_offset_best(sparse, dense, None, (0, 1.05), (0.05, None)))
    # back-projection (utils/create_point_cloud_from_touches.py:19-73)
    depth = np.zeros((4, 6))
    depth[1, 2], depth[2, 4] = 1.0, 2.0
    color = (np.arange(4 * 6 * 3).reshape(4, 6, 3) % 255).astype(np.float64)
    T = np.eye(4)
    T[:3, 3] = [1, 2, 3]
    pts, cols = cpc.get_point_cloud_from_depth_and_color(depth, color, [100, 100, 3, 2], T)
    out["bp/depth"], out["bp/color"], out["bp/T"], out["bp/points"], out["bp/colors"] = depth, color, T, pts, cols
    th = 0.7
    T2 = np.array([[np.cos(th), 0, np.sin(th), 0.3], [0, 1, 0, -0.2], [-np.sin(th), 0, np.cos(th), 1.0], [0, 0, 0, 1]])
    rng = np.random.default_rng(9)
    depth2 = np.where(rng.random((12, 16)) < 0.4, 0.5 + rng.random((12, 16)), 0.0)
    color2 = rng.integers(0, 255, (12, 16, 3)).astype(np.float64)
    pts2, cols2 = cpc.get_point_cloud_from_depth_and_color(depth2, color2, [80.0, 75.0, 7.5, 6.2], T2)
    out["bp2/depth"], out["bp2/color"], out["bp2/T"], out["bp2/points"], out["bp2/colors"] = depth2, color2, T2, pts2, cols2
    # train/eval split (utils/create_point_cloud_from_touches.py:174-198)
    for n, f in ((151, 0.08), (100, 0.13), (40, 0.8), (17, 0.5)):
        tr, ev = cpc.get_train_eval_split_fraction(list(range(n)), f)
        out[f"split/{n}_{f}/train"], out[f"split/{n}_{f}/eval"] = np.asarray(tr), np.asarray(ev)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fusion_reference.npz")
    np.savez_compressed(path, **{k: (v.astype(np.float32) if (v.dtype == np.float64 and v.shape == (H, W)) else v)
                                 for k, v in out.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
