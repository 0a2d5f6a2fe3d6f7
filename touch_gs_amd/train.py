"""Self-contained trainer for the ``depth-gaussian-splatting`` method (the build's counterpart of
``ns-train`` / ``ns-eval``, which never travel to the GPU box).

Flags keep the reference's names (scripts/train_bunny_real.sh:52, train_block_data.sh:50):
    python -m touch_gs_amd.train --data <scene> --depth-loss-mult 0.2 \
        --depth-loss-type DEPTH_UNCERTAINTY_WEIGHTED_LOSS --uncertainty-weight 1 \
        --train-split-fraction 0.8 [--max-num-iterations 30000] [--output-dir outputs]
    python -m torch.distributed.run --nproc-per-node 8 -m touch_gs_amd.train ...   # data parallel
    python -m touch_gs_amd.train --synthetic 100000 800 800 ...                    # no data needed

Outputs ``<output-dir>/<scene>/depth-gaussian-splatting/<timestamp>/`` with config.json, checkpoints
``step-<n>.ckpt`` (every --steps-per-save) and ``eval.json`` whose ``results`` keys are the ones the
reference aggregates (experiment_utils/get_results.py:35-52: psnr, ssim, depth_mse,
supervised_depth_mse; lpips is written as NaN -- it needs pretrained network weights).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import math
import os
import time

import torch

from . import ops, parallel
from .model import DEPTH_LOSS_TYPES, DepthGaussianSplattingModel, ModelConfig, View
from .optim import GaussianParams

SH_C0 = 0.28209479177387814


def init_params(n: int, K: int, device, seed_points=None, extent: float = 1.0, seed: int = 0,
                seed_fraction: float = 0.5) -> GaussianParams:
    """``n`` initial Gaussians: the touch point cloud (points_touch.npy -- the object's touched surface,
    utils/create_point_cloud_from_touches.py:243-244) + a uniform random fill of the cube [-extent, extent]^3 for
    everything the fingers never reached (table, background), Splatfacto-style: scales from the mean distance to the
    3 nearest neighbours, random rotations, opacity 0.1, SH dc from the point colour.

    A touch cloud can hold far more points than ``n`` (every touch-depth pixel of every training view): it is then
    SUBSAMPLED uniformly at random to ``seed_fraction * n`` points, so that all views contribute and the rest of the
    scene still gets its random fill.  (Rounds 1-4 kept the first ``n`` points in file order -- the first views'
    touches only -- and dropped the fill whenever the cloud was larger than ``n``.)"""
    g = torch.Generator().manual_seed(seed)
    pts, cols = [], []
    if seed_points is not None and len(seed_points[0]):
        sp, sc = seed_points[0].float(), seed_points[1].float() / 255.0
        cap = max(int(n * seed_fraction), 1)
        if len(sp) > cap:
            sel = torch.randperm(len(sp), generator=g)[:cap]
            sp, sc = sp[sel], sc[sel]
        pts.append(sp)
        cols.append(sc)
    n_rand = max(n - sum(len(p) for p in pts), 0)
    if n_rand:
        pts.append((torch.rand(n_rand, 3, generator=g) - 0.5) * 2 * extent)
        cols.append(torch.rand(n_rand, 3, generator=g))
    means = torch.cat(pts)[:n].float()
    colors = torch.cat(cols)[:n].float()
    N = means.shape[0]
    # mean distance to the 3 nearest neighbours, exact, in blocks (the cloud mixes a dense object with a sparse fill:
    # a subsampled estimate would give the object's points the fill's spacing)
    md = means.to(device)
    blk = max(64, min(4096, (1 << 28) // max(N, 1)))   # <= 1 GiB of distances per block whatever N is
    kd = torch.empty(N, device=md.device)
    for b in range(0, N, blk):
        d = torch.cdist(md[b:b + blk], md)
        kd[b:b + blk] = d.topk(4, largest=False).values[:, 1:].mean(1).clamp_min(1e-5)
    knn = kd.cpu()
    scale = knn
    sh = torch.zeros(N, K, 3)
    sh[:, 0] = (colors - 0.5) / SH_C0
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
    return GaussianParams.from_tensors(means.to(device), torch.log(scale)[:, None].repeat(1, 3).to(device),
                                       quats.to(device), torch.full((N,), math.log(0.1 / 0.9)).to(device),
                                       sh.to(device))


@torch.no_grad()
def evaluate(model: DepthGaussianSplattingModel, views) -> dict:
    acc = {}
    for v in views:
        out = model.get_outputs(v.cam, sh_degree=model.active_sh_degree())
        m, _ = model.get_image_metrics_and_images(out, v)
        for k, x in m.items():
            acc.setdefault(k, []).append(x)
    return {k: float(sum(x) / len(x)) for k, x in acc.items()}


def render_views(model: DepthGaussianSplattingModel, views, out_dir: str, names=None) -> None:
    """The build's counterpart of ``ns-render dataset`` (reference experiment_utils/run_eval.py:48):
    ``<out_dir>/rgb/<name>.png`` (8-bit) and ``<out_dir>/depth/<name>.png`` (uint16 millimetres, the
    convention of the dataset's own depth maps: utils/fuse_touch_vision.py:372-376)."""
    from PIL import Image
    from .plumbing import write_png16
    os.makedirs(os.path.join(out_dir, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "depth"), exist_ok=True)
    for i, v in enumerate(views):
        out = model.get_outputs(v.cam, sh_degree=model.active_sh_degree())
        name = os.path.splitext(os.path.basename(names[i]))[0] if names else f"{i:05d}"
        rgb = (out["rgb"].clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).cpu().numpy()
        Image.fromarray(rgb).save(os.path.join(out_dir, "rgb", name + ".png"))
        mm = (out["depth"].reshape(v.cam.H, v.cam.W) * 1000.0).clamp(0, 65535).round().to(torch.int32).cpu().numpy()
        write_png16(os.path.join(out_dir, "depth", name + ".png"), mm.astype("uint16"))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", type=str, default=None)
    ap.add_argument("--synthetic", type=int, nargs=3, metavar=("N", "W", "H"), default=None)
    ap.add_argument("--depth-loss-mult", type=float, default=0.2)
    ap.add_argument("--depth-loss-type", type=str, default="DEPTH_UNCERTAINTY_WEIGHTED_LOSS", choices=DEPTH_LOSS_TYPES)
    ap.add_argument("--uncertainty-weight", "--uncertainty_weight", type=float, default=1.0, dest="uncertainty_weight")
    ap.add_argument("--train-split-fraction", type=float, default=0.9)
    ap.add_argument("--uncertainty-scaling", type=str, default="linear", choices=("linear", "variance", "none"),
                    help="units of the uncertainty map in the scaled scene (dataset.py docstring; UNVERIFIED-PRIOR)")
    ap.add_argument("--uncertainty-floor", type=float, default=None,
                    help="lower bound of the uncertainty map, in the map's own units (dataset.py docstring: without it "
                         "the touched pixels pile the refinement onto the object); default 0 = off = the reference's loss "
                         "for the reference's flags (0.05 under --preset few-view)")
    ap.add_argument("--cull-unseen", dest="cull_unseen", action="store_true", default=None,
                    help="cull Gaussians no training view has had in its frustum for a whole refinement window "
                         "(densify.py; not in Splatfacto; on under --preset few-view)")
    ap.add_argument("--preset", type=str, default="reference", choices=("reference", "few-view"),
                    help="reference: the reference's flags mean what they mean there (uncertainty floor 0, no unseen cull). "
                         "few-view: the two additions that make touch-cloud seeds + a random fill train in the 8-13 view "
                         "regime (DESIGN.md section 10: --uncertainty-floor 0.05, --cull-unseen); explicit flags win")
    ap.add_argument("--max-num-iterations", type=int, default=30000)
    ap.add_argument("--steps-per-save", type=int, default=2000)
    ap.add_argument("--steps-per-eval", type=int, default=500)
    ap.add_argument("--num-gaussians", type=int, default=100000)
    ap.add_argument("--random-extent", type=float, default=1.0,
                    help="half side of the cube the random fill of the initial Gaussians is drawn from (scaled scene units)")
    ap.add_argument("--seed-fraction", type=float, default=0.5,
                    help="largest share of the initial Gaussians taken from the touch point cloud")
    ap.add_argument("--eval-views-during-training", type=int, default=0,
                    help="at every --steps-per-eval also print held-out PSNR / depth MSE over this many eval views")
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--sync-budget", action="store_true",
                    help="read the intersection count back every step instead of the sync-free speculative budget")
    ap.add_argument("--render-output", type=str, default=None,
                    help="also dump the eval views' renders (rgb/ + 16-bit mm depth/), like ns-render dataset")
    ap.add_argument("--output-dir", type=str, default="outputs")
    ap.add_argument("--load-checkpoint", type=str, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-spatial-sort", action="store_true",
                    help="keep the Gaussians in their initial row order instead of 3-D Morton order")
    ap.add_argument("--densify", dest="densify", action="store_true", default=True,
                    help="Splatfacto-style clone/split/cull refinement (default: on, as in the method the reference trains)")
    ap.add_argument("--no-densify", dest="densify", action="store_false", help="fixed set of Gaussians")
    ap.add_argument("--num-downscales", type=int, default=2,
                    help="Splatfacto's coarse-to-fine schedule: start at 1/2^n resolution (default 2; 0 = off)")
    ap.add_argument("--resolution-schedule", type=int, default=250, help="double the resolution every this many steps")
    ap.add_argument("--refine-every", type=int, default=100)
    ap.add_argument("--warmup-length", type=int, default=500)
    ap.add_argument("--densify-grad-thresh", type=float, default=0.0002)
    args = ap.parse_args(argv)

    few = args.preset == "few-view"
    if args.uncertainty_floor is None:
        args.uncertainty_floor = 0.05 if few else 0.0
    if args.cull_unseen is None:
        args.cull_unseen = few
    dp = parallel.init_from_env()
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    K = (args.sh_degree + 1) ** 2
    if args.synthetic:
        from .scene import make_view, synthetic_gaussians
        N, W, H = args.synthetic
        views = [make_view(N, W, H, args.sh_degree, 1235, dev, view=v, n_views=8) for v in range(8)]
        i_train, i_eval = list(range(7)), [7]
        P, _ = synthetic_gaussians(N, W, H, args.sh_degree, 4321)  # a *different* scene to fit from
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        scene_name = f"synthetic_{N}_{W}x{H}"
    else:
        from .dataset import Scene
        scene = Scene(args.data, args.train_split_fraction, dev, uncertainty_scaling=args.uncertainty_scaling,
                      uncertainty_floor=args.uncertainty_floor)
        views, i_train, i_eval = scene.views, list(scene.i_train), list(scene.i_eval)
        params = init_params(args.num_gaussians, K, dev, scene.seed_points(), extent=args.random_extent, seed=args.seed,
                             seed_fraction=args.seed_fraction)
        scene_name = os.path.basename(os.path.normpath(args.data))
    cfg = ModelConfig(sh_degree=args.sh_degree, depth_loss_mult=args.depth_loss_mult,
                      depth_loss_type=args.depth_loss_type, uncertainty_weight=args.uncertainty_weight,
                      spatial_sort=not args.no_spatial_sort, num_downscales=args.num_downscales,
                      resolution_schedule=args.resolution_schedule)
    model = DepthGaussianSplattingModel(cfg, params)
    if cfg.spatial_sort:
        model.spatial_sort()
    if args.densify:
        from .densify import DensifyConfig
        model.enable_densification(DensifyConfig(refine_every=args.refine_every, warmup_length=args.warmup_length,
                                                 num_train_data=len(i_train), densify_grad_thresh=args.densify_grad_thresh,
                                                 cull_unseen=bool(args.cull_unseen)))
    trainer_state = dict(uncertainty_scaling=args.uncertainty_scaling, uncertainty_floor=args.uncertainty_floor, densify=bool(args.densify),
                         num_downscales=args.num_downscales, cull_unseen=bool(args.cull_unseen))
    if dp.rank == 0 and (args.uncertainty_floor > 0 or args.cull_unseen):
        print(f"note: this run deviates from the reference's objective / Splatfacto's refinement: uncertainty floor "
              f"{args.uncertainty_floor} (no depth weight above 1 / (uncertainty_weight * floor)), cull_unseen={bool(args.cull_unseen)} "
              f"(--preset {args.preset}; DESIGN.md section 10)", flush=True)
    if args.load_checkpoint:
        sd = torch.load(args.load_checkpoint, map_location=dev)
        model.load_state_dict(sd)
        # the defaults of these three changed between rounds (uncertainty maps linear instead of squared, densification
        # and the resolution schedule on by default): resuming under different settings silently trains a different
        # objective, so the checkpoint records them and a mismatch is reported
        was = sd.get("trainer")
        if was is None:
            print("warning: the checkpoint predates the recorded trainer settings; this run uses "
                  f"{trainer_state} -- pass the flags of the original run explicitly if they differed", flush=True)
        else:
            for k, v in trainer_state.items():
                if k in was and was[k] != v:
                    print(f"warning: checkpoint was trained with {k}={was[k]!r}, this run uses {k}={v!r}", flush=True)
    run_dir = os.path.join(args.output_dir, scene_name, "depth-gaussian-splatting", time.strftime("%Y-%m-%d_%H%M%S"))
    if dp.rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, "config.json"), "w") as f:
            json.dump(dict(vars(args), model=dataclasses.asdict(cfg), world_size=dp.world,
                           scene=None if args.synthetic else scene.describe()), f, indent=2)
    if not args.sync_budget and (dp.world == 1 or model.optimizer.can_gather_sh()):
        # no per-step host sync; an overflow is detected late and replayed (data parallel: the ranks
        # agree on it on the device and replay the same steps)
        # TGS_SPEC_CAPACITY: test hook -- an initial capacity far too small forces the overflow / replay path
        model.enable_speculative_budget(capacity=int(os.environ.get("TGS_SPEC_CAPACITY", "0")))
    train_views = [views[i] for i in i_train]
    eval_views = [views[i] for i in i_eval] or train_views[:1]
    t0 = time.time()
    seen_refine, n_refines = None, 0
    for step in range(model.step, args.max_num_iterations):
        view = train_views[dp.views_for_step(step, len(train_views))]
        # single process: tell the step which view follows (colour prefetch, model.train_step)
        nxt = train_views[dp.views_for_step(step + 1, len(train_views))]   # this rank's next view (prefetches)
        model.train_step(view, dp if dp.active else None, next_view=nxt)
        if getattr(model, "_refined_at", None) == model.step and seen_refine != model.step:
            seen_refine, n_refines = model.step, n_refines + 1
        at_eval = (step + 1) % args.steps_per_eval == 0
        at_save = (step + 1) % args.steps_per_save == 0 or step + 1 == args.max_num_iterations
        if at_eval or at_save:
            # EVERY rank drains its pending overflow verdicts at the same steps: a drain that finds an agreed
            # overflow replays steps, collectives included, so a rank-0-only flush would leave the ranks'
            # collectives unmatched (and break "every rank inspects step s - L at step s")
            model.flush()
            dp.check_transport()      # peer transport: a wait that timed out (a rank never delivered) raises here
        if dp.rank == 0 and at_eval:
            loss = model.loss_from(model.last["tile_loss"], model.last["ssim_sum"], model.last["view"])
            print(f"step {step + 1}: " + " ".join(f"{k}={float(v):.5f}" for k, v in loss.items()) +
                  f"  N={model.params.N}  {(step + 1 - 0) / (time.time() - t0):.1f} it/s", flush=True)
            if args.eval_views_during_training > 0:
                k = max(len(eval_views) // args.eval_views_during_training, 1)
                r = evaluate(model, eval_views[::k][:args.eval_views_during_training])
                print(f"   held-out: " + " ".join(f"{a}={b:.5g}" for a, b in r.items()), flush=True)
        if dp.rank == 0 and at_save:
            torch.save(dict(model.state_dict(), trainer=trainer_state), os.path.join(run_dir, f"step-{step + 1:09d}.ckpt"))
    model.flush()
    if dp.world > 1:
        dp.assert_replicas_identical(model.params.flat)
    if dp.rank == 0:
        print(f"gaussians {model.params.N}  refinements {n_refines}  speculative replays "
              f"{getattr(model, 'speculative_replays', 0)}", flush=True)
    if dp.rank == 0:
        results = evaluate(model, eval_views)
        # the reference's aggregator indexes results['lpips'] unconditionally
        # (experiment_utils/get_results.py:38); LPIPS needs pretrained network weights that cannot be
        # fetched here, so the key is present and NaN rather than absent
        results.setdefault("lpips", float("nan"))
        with open(os.path.join(run_dir, "eval.json"), "w") as f:
            json.dump({"experiment_name": scene_name, "method_name": "depth-gaussian-splatting",
                       "checkpoint": run_dir, "results": results}, f, indent=2)
        print(json.dumps(results))
        if args.render_output:
            render_views(model, eval_views, args.render_output,
                         None if args.synthetic is not None else [scene.names[i] for i in i_eval] if i_eval else None)
    dp.barrier()
    if dp.peer is not None:
        dp.peer.close()
        dp.peer = None
    return run_dir


if __name__ == "__main__":
    main()
