"""The build's counterpart of the reference's ``experiment_utils/run_eval.py`` (which shells out to
``ns-eval`` and ``ns-render dataset``, :37-57): walks the run directories of one scene newest first,
evaluates the last checkpoint of each and writes

    <output_dir>/<exp_name>/<exp_name>_<k>.json     k = 1 .. past_n_trials   (run_eval.py:40-43)
    <exp_name>_renders/                             eval-view renders        (run_eval.py:48)

with the ``results`` keys the reference's aggregator reads (experiment_utils/get_results.py:35-52:
psnr, ssim, lpips (NaN: needs pretrained weights), depth_mse, supervised_depth_mse and -- when
IS_REAL_WORLD is exported as scripts/train_bunny_real.sh:54 does -- gt_depth_mse, gt_object_depth_mse).

    IS_REAL_WORLD=True python -m touch_gs_amd.run_eval --input_dir outputs/<scene>/depth-gaussian-splatting \\
        --output_dir experiments --exp_name bunny_real_exp --past_n_trials 1
"""
from __future__ import annotations

import argparse
import glob
import json
import os

import torch


def eval_run(run_dir: str, out_json: str, render_dir=None, device="cuda") -> dict:
    """Rebuild the model of one training run from its config.json + newest checkpoint and evaluate
    it on the run's eval split."""
    from .dataset import Scene
    from .model import DepthGaussianSplattingModel, ModelConfig
    from .optim import GaussianParams
    from .train import evaluate, render_views
    with open(os.path.join(run_dir, "config.json")) as f:
        cfg = json.load(f)
    ckpts = sorted(glob.glob(os.path.join(run_dir, "step-*.ckpt")))
    if not ckpts:
        raise FileNotFoundError(f"no checkpoint in {run_dir}")
    sd = torch.load(ckpts[-1], map_location=device)
    mc = {k: v for k, v in cfg["model"].items() if k in ModelConfig.__dataclass_fields__}
    mc["background_color"] = tuple(mc.get("background_color", (0.0, 0.0, 0.0)))
    model = DepthGaussianSplattingModel(ModelConfig(**mc), GaussianParams.allocate(sd["N"], sd["K"], device))
    model.load_state_dict(sd)
    if cfg.get("synthetic"):
        from .scene import make_view
        N, W, H = cfg["synthetic"]
        views = [make_view(N, W, H, cfg["sh_degree"], 1235, device, view=7, n_views=8)]
        names = None
    else:
        scene = Scene(cfg["data"], cfg["train_split_fraction"], device)
        idx = list(scene.i_eval) or list(scene.i_train)[:1]
        views, names = [scene.views[i] for i in idx], [scene.names[i] for i in idx]
    results = evaluate(model, views)
    results.setdefault("lpips", float("nan"))   # get_results.py:38 indexes it unconditionally
    os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
    with open(out_json, "w") as f:
        json.dump({"experiment_name": os.path.basename(os.path.dirname(os.path.dirname(run_dir))),
                   "method_name": "depth-gaussian-splatting", "checkpoint": ckpts[-1], "results": results}, f, indent=2)
    if render_dir:
        render_views(model, views, render_dir, names)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="Run evaluation on the output directories.")
    ap.add_argument("--input_dir", required=True, help="outputs/<scene>/depth-gaussian-splatting")
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--exp_name", required=True)
    ap.add_argument("--past_n_trials", type=int, required=True)
    ap.add_argument("--no_render", action="store_true")
    a = ap.parse_args(argv)
    full_exp_dir = os.path.join(a.output_dir, a.exp_name)
    os.makedirs(full_exp_dir, exist_ok=True)
    done = []
    for run in sorted(os.listdir(a.input_dir))[::-1]:            # newest first (timestamped names)
        run_dir = os.path.join(a.input_dir, run)
        if not os.path.exists(os.path.join(run_dir, "config.json")):
            continue
        out_json = os.path.join(full_exp_dir, f"{a.exp_name}_{len(done) + 1}.json")
        res = eval_run(run_dir, out_json, None if a.no_render else f"{a.exp_name}_renders")
        print(out_json, json.dumps(res))
        done.append(out_json)
        if len(done) == a.past_n_trials:
            break
    return done


if __name__ == "__main__":
    main()
