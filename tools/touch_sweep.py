"""Developer tool: the few-view run of the reduced quality pipeline with 25 / 50 / 100 touches (+ the
RGB-only run once) and, on the 50-touch capture, the dense flag set under the trainer's reference-parity defaults
(--preset reference: uncertainty floor 0, no unseen cull) next to the few-view preset.
    python tools/touch_sweep.py [--out gpurun_out/r6_touch_sweep.json]"""
import argparse, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from touch_gs_amd import analytic_scene as A
ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/r6_touch_sweep.json")
ap.add_argument("--iters", type=int, default=4000)
a = ap.parse_args()
keys = ("psnr", "ssim", "depth_mse", "gt_depth_mse", "gt_object_depth_mse", "gt_depth_mse_true_object_mask", "exact_depth_mse",
        "exact_object_depth_mse", "exact_object_depth_median_abs_m", "iters_per_s_wall")
out = {"size": "24 views 640x360, %d iterations, few-view split 0.25" % a.iters, "touches": {}}
for n in (25, 50, 100):
    root = tempfile.mkdtemp(prefix=f"ts{n}_")
    r = A.quick_quality(root, iters=a.iters, n_touches=n, runs=("bunny_real:1", "bunny_real:0") if n == 50 else ("bunny_real:1",))
    out["touches"][n] = {"gpis_object_cover": r["capture"]["gpis_object_cover"], "gpis_rmse_m": r["capture"]["gpis_rmse_m"],
                         "runs": {k: {m: v[m] for m in keys if m in v} for k, v in r["runs"].items()}}
    if n == 50:   # the dense flag set under both presets on the same capture
        for preset in ("few-view", "reference"):
            rr = A.train_and_eval(root, "block", True, iters=a.iters, preset=preset, num_gaussians=50000, extra_args=["--steps-per-eval", str(a.iters)])
            out.setdefault("block_0p8_split_presets", {})[preset] = {m: rr[m] for m in keys + ("gaussian_count",) if m in rr}
    print(n, json.dumps(out["touches"][n]), flush=True)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out))
