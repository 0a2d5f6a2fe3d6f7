"""Generates tests/golden/raster_selforacle.npz: tiny scenes rendered by the fp64 torch oracle.

SELF-ORACLE fixtures (NOT reference outputs): the reference tree contains no rasterizer to run
(oracle/__init__.py).  They pin the oracle against accidental drift and give the GPU tests a
fixed known-answer set.  Run from the repo root:  python tests/golden/make_raster_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import torch_oracle as O  # noqa: E402

CASES = [("one", 1, 32, 32, 0, 101), ("seven", 7, 48, 80, 2, 102), ("two_hundred", 200, 80, 48, 3, 103)]


def main():
    out = {}
    for name, N, W, H, deg, seed in CASES:
        P, c = O.synthetic_scene(N, W, H, deg, seed)
        if N == 1:  # a single centred Gaussian: closed-form sanity case
            P["means"][:] = torch.tensor([[0.05, -0.03, 3.0]], dtype=torch.float64)
            P["log_scales"][:] = torch.tensor([[-2.0, -2.5, -2.2]], dtype=torch.float64)
        cam = O.Camera(viewmat=O.orbit_viewmat(1 if N > 1 else 0, 8), **c, bg=(0.1, 0.2, 0.3))
        for k in P:
            P[k].requires_grad_(True)
        res, pr, gid, ts = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, deg)
        g = torch.Generator().manual_seed(seed)
        wr = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
        wd = torch.randn(H, W, generator=g, dtype=torch.float64)
        wa = torch.randn(H, W, generator=g, dtype=torch.float64)
        L = (res["rgb"] * wr).sum() + (res["depth_acc"] * wd).sum() + (res["alpha"] * wa).sum()
        L.backward()
        out[f"{name}/meta"] = np.array([N, W, H, deg, seed], dtype=np.int64)
        out[f"{name}/intr"] = np.array([c["fx"], c["fy"], c["cx"], c["cy"]], dtype=np.float64)
        out[f"{name}/viewmat"] = cam.viewmat.numpy()
        out[f"{name}/bg"] = np.array(cam.bg)
        for k in P:
            out[f"{name}/in/{k}"] = P[k].detach().numpy()
            out[f"{name}/grad/{k}"] = P[k].grad.numpy()
        for k in ("rgb", "depth_acc", "alpha"):
            out[f"{name}/out/{k}"] = res[k].detach().numpy().astype(np.float32)
        out[f"{name}/w_rgb"], out[f"{name}/w_depth"], out[f"{name}/w_alpha"] = wr.numpy(), wd.numpy(), wa.numpy()
        out[f"{name}/sorted_gid"] = gid
        out[f"{name}/tile_start"] = ts
    path = os.path.join(os.path.dirname(__file__), "raster_selforacle.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
