"""Developer tool: the trainer's steady-state step loop on a SAVED converged model (the reference's regime: 1280x720,
object on a table, 80 - 150 k Gaussians), for rocprofv3 passes and per-kernel event timings.

    TQ_SAVE_MODEL=1 python tools/train_quality.py --runs bunny_real:1,block:1 --images gpurun_out/ckpt   # writes model_*.pt
    python tools/ckpt_loop.py build_ab/ckpt/model_bunny_real_1.pt [--steps 200] [--warmup 30] [--breakdown] [--json out.json]

The checkpoint holds the flat parameter buffer and every camera of the capture; the supervision images are the model's
own renders from the training cameras (the capture's images do not travel: 64 MiB limit), slightly perturbed so that the
loss and its gradients are not identically zero.  Kernel times do not depend on the image values.  The loop is the
trainer's: speculative (sync-free) budget, next view announced (colour + front prefetch), fused K8 + Adam."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from touch_gs_amd import ops
from touch_gs_amd.camera import Camera
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig, View
from touch_gs_amd.optim import GaussianParams


def load(path, dev, n_views=0, freeze=True):
    sd = torch.load(path, map_location="cpu")
    mc = dict(sd["model"])
    mc["background_color"] = tuple(mc["background_color"])
    mc["num_downscales"] = 0            # steady state: full resolution
    if freeze:                          # keep the converged model where it is while it is being timed
        for k in ("lr_means", "lr_scales", "lr_quats", "lr_opac", "lr_sh_dc", "lr_sh_rest"):
            mc[k] = 0.0
        mc["lr_means_final"] = None
    params = GaussianParams.allocate(sd["N"], sd["K"], dev)
    params.flat.copy_(sd["flat"].to(dev))
    m = DepthGaussianSplattingModel(ModelConfig(**mc), params)
    m.step = int(sd["step"])
    ev = set(sd["i_eval"])
    idx = [i for i in range(len(sd["cams"])) if i not in ev]
    if n_views:
        idx = idx[:: max(len(idx) // n_views, 1)][:n_views]
    g = torch.Generator(device="cpu").manual_seed(1)
    views = []
    for i in idx:
        c = sd["cams"][i]
        cam = Camera(c["viewmat"], c["fx"], c["fy"], c["cx"], c["cy"], c["W"], c["H"], bg=mc["background_color"])
        with torch.no_grad():
            o = m.get_outputs(cam, sh_degree=m.active_sh_degree())
        H, W = cam.H, cam.W
        rgb = (o["rgb"] + 0.02 * torch.randn(H, W, 3, generator=g).to(dev)).clamp(0, 1).contiguous()
        d = o["depth"][..., 0]
        depth = torch.where(o["alpha"] > 0.5, d * (1 + 0.01 * torch.randn(H, W, generator=g).to(dev)), torch.zeros_like(d)).contiguous()
        unc = (0.05 + 0.2 * torch.rand(H, W, generator=g)).to(dev)
        v = View(cam=cam, rgb=rgb, depth=depth, uncertainty=unc)
        v.valid_count()
        views.append(v)
    return m, views


def breakdown(m, views, reps=5):
    """Per-op time of a train step, sync between ops (no overlap): the table of tools/train_quality.py --breakdown."""
    p_, c_ = m.params, m.config
    rows = {}

    def T(name, fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r_ = fn()
        e1.record(); torch.cuda.synchronize()
        rows.setdefault(name, []).append(round(e0.elapsed_time(e1) / reps * 1e3, 1))
        return r_
    for v in views:
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
        rect = splats.view(-1, 12)[:, 10].view(torch.int32)
        hits = ((rect >> 16) & 255) * ((rect >> 24) & 255)
        rows.setdefault("max_hits", []).append(int(hits.max()))
        rows.setdefault("gaussians_over_32_tiles", []).append(int((hits > 32).sum()))
        rows.setdefault("pairs_in_those", []).append(int(hits[hits > 32].sum()))
        gh = torch.nn.functional.pad(hits, (0, (-len(hits)) % 256)).view(-1, 256).sum(1)
        rows.setdefault("largest_group_pairs", []).append(int(gh.max()))
        rgb, dacc, fT, _ = T("k6_us", lambda: ops.rasterize_fwd(cam, splats, sg, ts))
        ss, vimg = T("ssim_us", lambda: ops.ssim_fwd_bwd(rgb, v.rgb, weight=-c_.ssim_lambda / (3 * H * W), reduce=False))
        parts, tl = T("k7_us", lambda: ops.rasterize_bwd(cam, splats, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=m.loss_spec(v), want_tile_loss=True))
        T("k8_us", lambda: ops.project_bwd(cam, p_.means, p_.log_scales, p_.quats, p_.opac_logit, p_.sh, deg, splats, gb, parts, out=p_.grad_views(), want_v_xy=True))
        T("adam_us", lambda: m.optimizer.step())
    rows["N"] = p_.N
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ckpt")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--views", type=int, default=0, help="use this many of the training views (0 = all)")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--no-freeze", action="store_true", help="keep the learning rates (the model then drifts from its optimum)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--layout", default="balanced", choices=("balanced", "morton"),
                    help="row order: Morton with the long-run Gaussians dealt over the groups (the trainer's) or plain Morton")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m, views = load(a.ckpt, dev, a.views, freeze=not a.no_freeze)
    out = {"ckpt": os.path.basename(a.ckpt), "N": m.params.N, "views": len(views), "W": views[0].cam.W, "H": views[0].cam.H}
    m.config.spatial_sort = True
    m.config.balance_long_runs = a.layout == "balanced"
    if a.layout == "balanced":     # the trainer re-sorts with steps behind it (it keeps their cameras): learning rates are frozen
        for v in views[:16]:
            m.train_step(v)
        torch.cuda.synchronize()
    m.spatial_sort()
    out["layout"] = a.layout
    out["long_run"] = ops.set_long_run()      # chosen by the re-sort (model.spatial_sort)
    if a.breakdown:
        out["breakdown"] = breakdown(m, views[:3])
    m.enable_speculative_budget()
    n = len(views)
    for s in range(a.warmup):
        m.train_step(views[s % n], next_view=views[(s + 1) % n])
    m.flush(); torch.cuda.synchronize()
    reps = []
    for r in range(3):
        t0 = time.perf_counter()
        for s in range(a.steps):
            m.train_step(views[s % n], next_view=views[(s + 1) % n])
        m.flush(); torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / a.steps * 1e3)
    out.update(ms_per_step=round(min(reps), 4), ms_per_step_repeats=[round(x, 4) for x in reps],
               iters_per_s=round(1e3 / min(reps), 1), replays=getattr(m, "speculative_replays", 0))
    print(json.dumps(out), flush=True)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
