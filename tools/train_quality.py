"""Does the trainer train?  ("Next" 1.)  The whole scripts/train_bunny_real.sh sequence on a known-geometry
capture: raw capture -> touch_gs_amd.prepare (RealSense re-intrinsic, GPIS maps -> touch depth, fake monocular depth ->
align -> fuse, transforms, seeds) -> touch_gs_amd.train for the full 30 000 iterations with the reference's two flag
sets (bunny_real: 0.08 split; block: 0.8 split), each WITH and WITHOUT the depth term -> run_eval.

    python tools/train_quality.py [--root DIR] [--views 100] [--iters 30000] [--runs bunny_real:1,bunny_real:0,block:1,block:0]

Writes gpurun_out/train_quality.json (+ every run's eval.json next to its checkpoint)."""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from touch_gs_amd import analytic_scene as A

ap = argparse.ArgumentParser()
ap.add_argument("--root", default=None)
ap.add_argument("--views", type=int, default=100)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--iters", type=int, default=30000)
ap.add_argument("--runs", default="bunny_real:1,bunny_real:0,block:1,block:0")
ap.add_argument("--num-gaussians", type=int, default=100000)
ap.add_argument("--out", default="gpurun_out/train_quality.json")
ap.add_argument("extra", nargs="*", help="extra trainer flags after --")
args = ap.parse_args()
root = args.root or tempfile.mkdtemp(prefix="tq_")
out = {"root": root}
if not os.path.exists(os.path.join(root, "transforms.json")):
    t = time.perf_counter()
    out["capture"] = A.write_raw_capture(root, n_views=args.views, W=args.width, H=args.width * 9 // 16, device="cuda")
    out["capture_s"] = round(time.perf_counter() - t, 1)
    print(json.dumps(out["capture"]), out["capture_s"], "s", flush=True)
if not os.path.exists(os.path.join(root, "fused_output_dir")):
    t = time.perf_counter()
    out["prepare"] = A.prepare_capture(root, 0.08)
    out["prepare_s"] = round(time.perf_counter() - t, 1)
    print(json.dumps(out["prepare"]), out["prepare_s"], "s", flush=True)
out["runs"] = {}
for spec in args.runs.split(","):
    flags, wd = spec.split(":")
    r = A.train_and_eval(root, flags, wd == "1", iters=args.iters, num_gaussians=args.num_gaussians, extra_args=args.extra)
    out["runs"][spec] = r
    print(spec, json.dumps(r), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
print(json.dumps(out))
