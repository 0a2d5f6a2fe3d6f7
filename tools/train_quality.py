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
ap.add_argument("--breakdown", action="store_true", help="with --images: per-op time of a train step of the final model")
ap.add_argument("--images", default=None, help="directory for [truth | render] pictures of two held-out views per run")
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
    flags, wd, *more = spec.split(":")     # bunny_real:1[:--flag+value+--flag+value]  (per-run trainer flags)
    r = A.train_and_eval(root, flags, wd == "1", iters=args.iters, num_gaussians=args.num_gaussians,
                         extra_args=list(args.extra) + (more[0].split("+") if more else []))
    out["runs"][spec] = r
    if args.images:   # held-out view 0: [truth | render] colour and depth, half size, for a look
        import glob, numpy as np
        from PIL import Image
        from touch_gs_amd.dataset import Scene
        from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
        from touch_gs_amd.optim import GaussianParams
        cfg = json.load(open(os.path.join(r["run_dir"], "config.json")))
        sd = torch.load(sorted(glob.glob(os.path.join(r["run_dir"], "step-*.ckpt")))[-1], map_location="cuda")
        mc = {k: v for k, v in cfg["model"].items() if k in ModelConfig.__dataclass_fields__}
        mc["background_color"] = tuple(mc["background_color"])
        m = DepthGaussianSplattingModel(ModelConfig(**mc), GaussianParams.allocate(sd["N"], sd["K"], "cuda"))
        m.load_state_dict(sd)
        scene = Scene(root, A.FLAG_SETS[flags]["split"], "cuda")
        os.makedirs(args.images, exist_ok=True)
        if os.environ.get("TQ_SAVE_MODEL"):
            torch.save(dict(flat=sd["flat"].cpu(), N=sd["N"], K=sd["K"], step=sd["step"], model=mc,
                            cams=[dict(viewmat=v.cam.viewmat.tolist(), fx=v.cam.fx, fy=v.cam.fy, cx=v.cam.cx, cy=v.cam.cy,
                                       W=v.cam.W, H=v.cam.H) for v in scene.views], i_eval=[int(i) for i in scene.i_eval]),
                       os.path.join(args.images, f"model_{flags}_{wd}.pt"))
        for j in list(scene.i_eval)[:: max(len(scene.i_eval) // 2, 1)][:2]:
            v = scene.views[j]
            o = m.get_outputs(v.cam, sh_degree=m.active_sh_degree())
            stem = os.path.splitext(os.path.basename(scene.names[j]))[0]
            gt = torch.from_numpy(np.load(os.path.join(root, "gt_depth", stem + ".npy"))).cuda().float() * scene.scale
            rgb = torch.cat([v.rgb, o["rgb"].clamp(0, 1)], 1)
            dep = (torch.cat([gt, o["depth"][..., 0], v.depth], 1) / 4.0).clamp(0, 1)[..., None].expand(-1, -1, 3)
            img = torch.cat([rgb, dep[:, :rgb.shape[1]]], 0)
            Image.fromarray((img * 255).to(torch.uint8).cpu().numpy()).resize((img.shape[1] // 2, img.shape[0] // 2)).save(
                os.path.join(args.images, f"{flags}_{wd}{'_x' if more else ''}_{stem}.jpg"), quality=85)
        if args.breakdown:   # where does a step of the FINAL model go?  (per-op, sync between ops: no overlap)
            from touch_gs_amd import ops
            tv = [scene.views[i] for i in list(scene.i_train)[:3]]
            p_, c_ = m.params, m.config
            rows = {}
            def T(name, fn, reps=5):
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    r_ = fn()
                e1.record(); torch.cuda.synchronize()
                rows.setdefault(name, []).append(round(e0.elapsed_time(e1) / reps * 1e3, 1))
                return r_
            for v in tv:
                cam, deg = v.cam, m.active_sh_degree()
                H, W = cam.H, cam.W
                b = ops.IntersectBudget()
                ops.project_bin_sort(cam, p_.means, p_.log_scales, p_.quats, p_.opac_logit, p_.sh, deg, b)
                b2 = ops.IntersectBudget(capacity=int(b.last_need * 1.2), sync=False)
                fr = T("front_us", lambda: ops.project_bin_sort(cam, p_.means, p_.log_scales, p_.quats, p_.opac_logit, p_.sh, deg, b2, want_radii=True))
                splats, radii, gb, ts, sg, st = fr
                lens = (ts[1:cam.num_tiles + 1] - ts[:cam.num_tiles])
                rows.setdefault("pairs", []).append(int(ts[cam.num_tiles])); rows.setdefault("longest_list", []).append(int(lens.max()))
                rows.setdefault("visible", []).append(int((radii > 0).sum()))
                rgb, dacc, fT, _ = T("k6_us", lambda: ops.rasterize_fwd(cam, splats, sg, ts))
                ss, vimg = T("ssim_us", lambda: ops.ssim_fwd_bwd(rgb, v.rgb, weight=-c_.ssim_lambda / (3 * H * W), reduce=False))
                parts, tl = T("k7_us", lambda: ops.rasterize_bwd(cam, splats, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=m.loss_spec(v), want_tile_loss=True))
                T("k8_us", lambda: ops.project_bwd(cam, p_.means, p_.log_scales, p_.quats, p_.opac_logit, p_.sh, deg, splats, gb, parts, out=p_.grad_views(), want_v_xy=True))
                T("adam_us", lambda: m.optimizer.step())
            rows["N"] = p_.N
            r["breakdown"] = rows
            print("breakdown", json.dumps(rows), flush=True)
        del m
    print(spec, json.dumps(r), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
print(json.dumps(out))
