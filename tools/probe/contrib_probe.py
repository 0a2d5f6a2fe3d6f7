"""Measures, on a BASELINE config, how many of the (pixel, Gaussian) evaluations of the compositing
kernels contribute (tools/probe/contrib_probe.hip).  Usage: python tools/probe/contrib_probe.py [cfg2|cfg3|cfg5]"""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
CFG = {"cfg2": (100_000, 800, 800, 3, 1235), "cfg3": (1_000_000, 1920, 1080, 3, 1236),
       "cfg5": (5_000_000, 3840, 2160, 3, 1238)}


def main():
    from touch_gs_amd import ops
    from touch_gs_amd._lib import ptr
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    so = os.path.join(HERE, "libprobe.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(HERE, "contrib_probe.hip"), "-o", so])
    lib = C.CDLL(so)
    lib.probe_contrib.argtypes = [C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 5
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    N, W, H, deg, seed = CFG[name]
    dev = torch.device("cuda:0")
    P, intr = synthetic_gaussians(N, W, H, deg, seed)
    D = {k: v.to(dev).contiguous() for k, v in P.items()}
    cam = make_camera(intr, 0, 8)
    sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
    ctr = torch.zeros(16, dtype=torch.int64, device=dev)
    rc = lib.probe_contrib(W, H, float(cam.pix_center), ptr(sp), ptr(sg), ptr(ts), ptr(ctr),
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    c = ctr.tolist()
    out = dict(config=name, pairs=c[0], quad_pairs_reaching_live=c[1], pixel_pairs_contributing=c[2],
               pixel_pairs_reach_but_dead_in_live_quad=c[3], blocks4x4_with_contribution=c[4],
               pairs_with_any_contribution=c[5], pairs_after_tile_death=c[6], rows8_with_contribution=c[7],
               quad_pairs_with_contribution=c[8], pixel_pairs_reaching=c[9],
               pairs_reaching_some_pixel=c[10])
    out["lane_utilisation_quadrant_slots"] = c[2] / max(64 * c[1], 1)
    out["lane_utilisation_if_4x4_blocks"] = c[2] / max(16 * c[4], 1)
    out["lane_utilisation_if_8px_rows"] = c[2] / max(8 * c[7], 1)
    out["quads_per_pair"] = c[1] / max(c[0], 1)
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"contrib_probe_{name}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
