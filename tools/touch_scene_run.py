"""The workload the reference actually runs: a scene at the reference's resolution that STARTS from a
few thousand touch seed points and GROWS under Splatfacto's refinement schedule.

    python tools/touch_scene_run.py [--steps 3000] [--disk]

* 1280 x 720 (reference utils/fuse_touch_vision.py:278), SH degree 3 with Splatfacto's ramp, 30 orbit views of an
  object-centric target scene (300 k Gaussians, 80 % of them in the central 10 % of the image), tactile depth +
  uncertainty supervision with the flags of scripts/train_block_data.sh:50;
* the model starts from 5 000 seed points drawn from the target's surface (what points_touch.npy is for) and refines
  with DensifyConfig defaults (warm-up 500, every 100 steps, opacity reset every 30 refinements) -- the trainer's
  defaults throughout: sync-free intersection budget, colour / front prefetch, Morton order, resolution schedule 2 / 250;
* reported per 500-step window: iters/s, N at the end, host time spent in refinement steps, replayed steps;
* small N is launch bound: eager launches against hipGraph replay of the same step at N = 5 000 (no refinement);
* --disk: the same through `python -m touch_gs_amd.train --data <dir>` on the scene written to disk in the
  reference's format (PNG decoding and uploads included in the wall time)."""
import argparse, json, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from touch_gs_amd import train
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.scene import write_scene_dir

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--disk", action="store_true")
ap.add_argument("--target", type=int, default=300_000)
ap.add_argument("--seeds", type=int, default=5000)
args = ap.parse_args()
dev = torch.device("cuda:0")
deg, NV = 3, 30
# ---- 1. the growing run (the function behind `train_touch_scene` of the bench line) ----
grow, views, (pts, cols) = bench.touch_scene_run(dev, steps=args.steps, target=args.target, seeds=args.seeds)
out = {"growing_run": grow}
print(json.dumps(grow), flush=True)


def fresh():
    params = train.init_params(args.seeds, 16, dev, (pts, cols), seed=0)
    cfg = ModelConfig(sh_degree=deg, sh_degree_interval=0, depth_loss_mult=0.2, depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS",
                      uncertainty_weight=1.0, spatial_sort=True)
    m = DepthGaussianSplattingModel(cfg, params)
    m.spatial_sort()
    return m


# ---- 2. small N: eager launches vs hipGraph replay (no refinement, full resolution) ----
res = {}
for mode in ("eager", "graphs"):
    m = fresh()
    if mode == "graphs":
        m.capture_step_graphs(views)
    else:
        m.enable_speculative_budget()
    for s in range(60):
        m.train_step(views[s % NV], next_view=None if mode == "graphs" else views[(s + 1) % NV])
    torch.cuda.synchronize(); t = time.perf_counter()
    for s in range(600):
        m.train_step(views[s % NV], next_view=None if mode == "graphs" else views[(s + 1) % NV])
    m.flush() if mode == "eager" else m.budget.check()
    torch.cuda.synchronize()
    res[mode] = round(600 / (time.perf_counter() - t), 1)
    del m
out["small_N_5000"] = {"iters_per_s": res, "graph_speedup": round(res["graphs"] / res["eager"], 3)}
print(json.dumps(out["small_N_5000"]), flush=True)
# ---- 3. the same through the trainer on an on-disk scene in the reference's format ----
if args.disk:
    root = tempfile.mkdtemp(prefix="touch_scene_")
    t = time.perf_counter()
    write_scene_dir(root, views, pts.numpy(), cols.numpy())
    t_write = time.perf_counter() - t
    t = time.perf_counter()
    run = train.main(["--data", root, "--train-split-fraction", "0.9", "--max-num-iterations", str(args.steps),
                      "--steps-per-eval", "1000", "--steps-per-save", str(args.steps), "--sh-degree", str(deg),
                      "--num-gaussians", str(args.seeds), "--output-dir", os.path.join(root, "out")])
    t_train = time.perf_counter() - t
    evj = json.load(open(os.path.join(run, "eval.json")))["results"]
    out["trainer_on_disk"] = {"write_scene_s": round(t_write, 1), "wall_s_incl_loading_eval_checkpoint": round(t_train, 2),
                              "iters_per_s_wall": round(args.steps / t_train, 1), "psnr": round(evj["psnr"], 2),
                              "gaussian_count": evj.get("gaussian_count")}
    print(json.dumps(out["trainer_on_disk"]), flush=True)
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/touch_scene.json", "w"), indent=1)
