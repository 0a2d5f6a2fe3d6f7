"""GPU tests of the drop-in operator surfaces: gsplat-0.1 shaped ops, INRIA-shaped
GaussianRasterizer, the Splatfacto-style model (autograd path == fused path) and the trainer."""
import math

import numpy as np
import time

import pytest
import torch

from oracle import torch_oracle as O
from tests.util import amd_cam, free_port, relerr, scene, to_dev

pytestmark = pytest.mark.gpu


def test_gsplat_shaped_ops_match_fused_path_and_oracle(dev):
    from touch_gs_amd import ops
    N, W, H, deg = 3000, 160, 96, 3
    P, cam = scene(N, W, H, deg, 101)
    acam = amd_cam(cam)
    D = {k: v.requires_grad_(True) for k, v in to_dev(P, dev).items()}
    viewmat = torch.from_numpy(cam.viewmat.numpy()).float().to(dev)
    # --- the way Splatfacto calls the ops ---
    scales = torch.exp(D["log_scales"])
    xys, depths, radii, conics, comp, num_tiles_hit, cov3d = ops.project_gaussians(
        D["means"], scales, 1.0, D["quats"], viewmat, cam.fx, cam.fy, cam.cx, cam.cy, H, W, 16)
    campos = torch.from_numpy(cam.campos().numpy()).float().to(dev)
    viewdirs = D["means"].detach() - campos
    rgbs = torch.clamp(ops.spherical_harmonics(deg, viewdirs, D["sh"]) + 0.5, min=0.0)
    bg = torch.tensor(cam.bg, device=dev)
    rgb, alpha = ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, rgbs, torch.sigmoid(D["opac_logit"])[:, None],
                                         H, W, 16, background=bg, return_alpha=True)
    depth_im = ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, depths[:, None].repeat(1, 3),
                                       torch.sigmoid(D["opac_logit"])[:, None], H, W, 16,
                                       background=torch.zeros(3, device=dev))[..., 0]
    assert cov3d.shape == (N, 6) and comp.shape == (N,) and num_tiles_hit.dtype == torch.int32
    # --- fused path on the same parameters ---
    D2 = {k: v.detach().clone().requires_grad_(True) for k, v in D.items()}
    rgb2, dacc2, alpha2, radii2 = ops.render(D2["means"], D2["log_scales"], D2["quats"], D2["opac_logit"], D2["sh"], acam, deg)
    assert torch.equal(radii, radii2)
    # the two-op path hands ABSOLUTE fp32 screen positions across the gsplat-shaped API (xys: 7.6e-6 px off the
    # fused path's rect-relative, compensated ones at x ~ 160): the images agree to 2e-5 except where that
    # difference flips an alpha >= 1/255 / T <= 1e-4 decision of a single pixel (seen: 1 pixel, 1.6e-4)
    d_rgb, d_a = (rgb - rgb2).abs().amax(-1), (alpha - alpha2).abs()
    assert int((d_rgb > 2e-5).sum()) <= 3 and int((d_a > 2e-5).sum()) <= 3, (int((d_rgb > 2e-5).sum()), int((d_a > 2e-5).sum()))
    assert d_rgb.max().item() < 2e-3 and d_a.max().item() < 2e-3
    assert int(((depth_im - dacc2).abs() > 2e-4 + 1e-4 * dacc2.abs()).sum()) <= 3
    # --- gradients through the two-op path equal the fused path's ---
    g = torch.Generator().manual_seed(3)
    w = torch.randn(H, W, 3, generator=g).to(dev)
    wd = torch.randn(H, W, generator=g).to(dev)
    ((rgb * w).sum() + (depth_im * wd).sum()).backward()
    ((rgb2 * w).sum() + (dacc2 * wd).sum()).backward()
    for k in D:
        if k == "means":  # the stand-alone SH op is fed detached view directions (as gsplat callers do)
            continue
        a, b = D[k].grad, D2[k].grad
        assert (a - b).abs().max().item() < 2e-3 * b.abs().max().item() + 1e-6, k
    # --- and the oracle ---
    out, *_ = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, deg)
    e = relerr(rgb.detach().cpu().numpy(), out["rgb"].detach().numpy(), floor=1e-2)
    assert np.quantile(e, 0.99) < 1e-4


def test_inria_shaped_rasterizer(dev):
    from touch_gs_amd import ops
    from touch_gs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    N, W, H, deg = 2000, 128, 80, 2
    P, cam = scene(N, W, H, deg, 111)
    D = to_dev(P, dev)
    V = torch.from_numpy(cam.viewmat.numpy()).float().to(dev)
    tanfovx, tanfovy = W / (2 * cam.fx), H / (2 * cam.fy)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy,
                                       bg=torch.tensor(cam.bg, device=dev), scale_modifier=1.0,
                                       viewmatrix=V.T.contiguous(), projmatrix=torch.eye(4, device=dev), sh_degree=deg,
                                       campos=torch.from_numpy(cam.campos().numpy()).float().to(dev))
    means = D["means"].clone().requires_grad_(True)
    means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
    opac = torch.sigmoid(D["opac_logit"])[:, None].clone().requires_grad_(True)
    scales = torch.exp(D["log_scales"]).clone().requires_grad_(True)
    rot = torch.nn.functional.normalize(D["quats"]).clone().requires_grad_(True)
    shs = D["sh"].clone().requires_grad_(True)
    color, radii, depth, alpha = GaussianRasterizer(rs, return_depth=True)(means, means2D, opac, shs=shs, scales=scales, rotations=rot)
    assert color.shape == (3, H, W) and radii.shape == (N,)
    out, pr, *_ = O.render(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, deg)
    e = relerr(color.permute(1, 2, 0).detach().cpu().numpy(), out["rgb"].detach().numpy(), floor=1e-2)
    assert np.quantile(e, 0.99) < 1e-4
    color.sum().backward()
    assert means2D.grad is not None and means2D.grad[:, :2].abs().sum() > 0 and float(means2D.grad[:, 2].abs().sum()) == 0
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in (means, opac, scales, rot, shs))
    # colours_precomp path == SH degree 0 with the same colours
    cols = torch.rand(N, 3, device=dev)
    c1, _ = GaussianRasterizer(rs)(means.detach(), None, opac.detach(), colors_precomp=cols, scales=scales.detach(), rotations=rot.detach())
    sh0 = ((cols - 0.5) / 0.28209479177387814)[:, None, :].contiguous()
    rs0 = rs._replace(sh_degree=0)
    c2, _ = GaussianRasterizer(rs0)(means.detach(), None, opac.detach(), shs=sh0, scales=scales.detach(), rotations=rot.detach())
    assert torch.allclose(c1, c2, atol=1e-6)
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(means, means2D, opac, scales=scales, rotations=rot)
    # cov3D_precomp (SURVEY 8(b) lists it in the signature): Sigma = R S^2 R^T given directly renders what the scales + rotations
    # render (the covariance is split back by eigh: other axes order / quaternion sign, same Gaussian), scale_modifier is
    # NOT applied to it (INRIA rule), and its gradient is finite and agrees with the directional derivative of the image sum
    w, x, y, z = rot.detach().unbind(1)
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    Sig = R @ torch.diag_embed(scales.detach() ** 2) @ R.transpose(1, 2)
    cov6 = torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1).requires_grad_(True)
    rs2 = rs._replace(scale_modifier=2.0)            # must not matter for a given covariance
    c3, r3 = GaussianRasterizer(rs2)(means.detach(), None, opac.detach(), shs=shs.detach(), cov3D_precomp=cov6)
    e = relerr(c3.permute(1, 2, 0).detach().cpu().numpy(), color.permute(1, 2, 0).detach().cpu().numpy(), floor=1e-2)
    assert np.quantile(e, 0.999) < 1e-4, np.quantile(e, 0.999)
    wgt = torch.rand_like(c3)
    (c3 * wgt).sum().backward()
    assert torch.isfinite(cov6.grad).all() and cov6.grad.abs().sum() > 0
    d = torch.randn_like(cov6) * Sig.diagonal(dim1=1, dim2=2).mean(1, keepdim=True) * 2e-3   # relative to each Gaussian's own size
    f = lambda c: float((GaussianRasterizer(rs2)(means.detach(), None, opac.detach(), shs=shs.detach(), cov3D_precomp=c)[0] * wgt).double().sum())
    fd = (f(cov6.detach() + d) - f(cov6.detach() - d)) / 2
    an = float((cov6.grad.double() * d.double()).sum())
    assert abs(fd - an) < 0.2 * abs(an) + 1e-2, (fd, an)     # (the image is only piecewise smooth in the covariance: 1/255 cut, tile rects)
    with pytest.raises(Exception):      # exactly one of (scales, rotations) / cov3D_precomp
        GaussianRasterizer(rs)(means, None, opac, shs=shs, scales=scales, rotations=rot, cov3D_precomp=cov6)


def test_model_autograd_path_equals_fused_path(dev):
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig, View
    from touch_gs_amd.optim import GaussianParams
    N, W, H, deg = 3000, 160, 96, 3
    P, cam = scene(N, W, H, deg, 121)
    g = torch.Generator().manual_seed(5)
    view = View(cam=amd_cam(cam), rgb=torch.rand(H, W, 3, generator=g).to(dev),
                depth=(torch.rand(H, W, generator=g) * 5 * (torch.rand(H, W, generator=g) > 0.3)).to(dev),
                uncertainty=(torch.rand(H, W, generator=g) * 5 + 1e-3).to(dev))
    for loss_type, mult, uw in (("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 0.2, 1.0), ("SIMPLE_LOSS", 0.5, 1.0),
                                ("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 0.005, 0.01)):  # the reference's three settings
        cfg = ModelConfig(sh_degree=deg, sh_degree_interval=0, depth_loss_mult=mult, depth_loss_type=loss_type,
                          uncertainty_weight=uw)
        params = GaussianParams.from_tensors(*[P[k].float().to(dev) for k in GaussianParams.NAMES])
        model = DepthGaussianSplattingModel(cfg, params)
        tl, ss = model.forward_backward(view)
        fused = {k: float(v) for k, v in model.loss_from(tl, ss, view).items()}
        fused_grads = {k: params.g[k].clone() for k in GaussianParams.NAMES}
        leaves = [getattr(params, k).detach().clone().requires_grad_(True) for k in GaussianParams.NAMES]
        p2 = GaussianParams.from_tensors(*[t.detach() for t in leaves])
        for k, t in zip(GaussianParams.NAMES, leaves):
            setattr(p2, k, t)
        m2 = DepthGaussianSplattingModel(cfg, p2)
        out = m2.get_outputs(view.cam)
        ld = m2.get_loss_dict(out, view)
        sum(ld.values()).backward()
        assert abs(float(ld["main_loss"]) - fused["main_loss"]) < 1e-4 * abs(fused["main_loss"])
        assert abs(float(ld["depth_loss"]) - fused["depth_loss"]) < 1e-4 * abs(fused["depth_loss"]) + 1e-9
        for k, t in zip(GaussianParams.NAMES, leaves):
            scale = fused_grads[k].abs().max().item()
            assert (t.grad - fused_grads[k]).abs().max().item() < 5e-4 * scale, (loss_type, k)
        md = m2.get_metrics_dict(out, view)
        assert "psnr" in md and "depth_mse" in md
        metrics, images = m2.get_image_metrics_and_images(out, view)
        assert {"psnr", "ssim", "depth_mse", "supervised_depth_mse"} <= set(metrics) and "img" in images


def test_training_reduces_loss_and_checkpoint_roundtrip(dev, tmp_path):
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 5000, 160, 96, 1
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0, lr_means=2e-3), params)
    losses = []
    for step in range(60):
        v = views[step % 4]
        model.train_step(v)
        if step % 4 == 0:
            losses.append(sum(float(x) for x in model.loss_from(model.last["tile_loss"], model.last["ssim_sum"], v).values()))
    assert losses[-1] < 0.9 * losses[0], losses
    sd = model.state_dict()
    torch.save(sd, tmp_path / "m.ckpt")
    p2 = GaussianParams.allocate(N, (deg + 1) ** 2, dev)
    m2 = DepthGaussianSplattingModel(model.config, p2)
    m2.load_state_dict(torch.load(tmp_path / "m.ckpt"))
    assert torch.equal(p2.flat, params.flat) and m2.step == model.step
    model.train_step(views[0]); m2.train_step(views[0])
    assert torch.equal(p2.flat, params.flat)  # deterministic kernels: identical continuation


def test_trainer_cli_synthetic(dev, tmp_path):
    from touch_gs_amd import train
    run = train.main(["--synthetic", "3000", "128", "80", "--max-num-iterations", "12", "--steps-per-eval", "6",
                      "--steps-per-save", "12", "--sh-degree", "1", "--output-dir", str(tmp_path),
                      "--render-output", str(tmp_path / "renders")])
    import json, os
    ev = json.load(open(os.path.join(run, "eval.json")))
    assert {"psnr", "ssim", "lpips", "depth_mse"} <= set(ev["results"]) and os.path.exists(os.path.join(run, "step-000000012.ckpt"))
    assert math.isnan(ev["results"]["lpips"])     # key present for the reference's aggregator, value unavailable
    # render dump (the ns-render counterpart): 8-bit rgb + 16-bit millimetre depth of the eval view
    from PIL import Image
    from touch_gs_amd.plumbing import read_png16
    rgb = np.asarray(Image.open(tmp_path / "renders" / "rgb" / "00000.png"))
    mm = read_png16(str(tmp_path / "renders" / "depth" / "00000.png"))
    assert rgb.shape == (80, 128, 3) and rgb.dtype == np.uint8 and mm.shape == (80, 128) and mm.dtype == np.uint16
    assert rgb.max() > 0 and mm.max() > 1000    # depths of the synthetic scene are metres -> thousands of mm


_DP_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dp = parallel.init_from_env(backend="gloo")           # 2 ranks share the single GPU; device tensors over gloo
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N, W, H, deg = 4001, 160, 96, 3
views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]
P, _ = synthetic_gaussians(N, W, H, deg, 99)
def fresh():
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
dp.n_chunks = 3
dp.color_chunks = 3          # pipelined factored exchange: K8 / gather / SH Adam in 3 row chunks
def train(factored):
    m = fresh()
    m.dp_factored_sh = factored
    for step in range(3):
        m.train_step(views[dp.views_for_step(step, 4)], dp)
    torch.cuda.synchronize()
    dp.assert_replicas_identical(m.params.flat)
    return m
model = train(True)       # colour-gradient all-gather + geometry all-reduce (default)
dense = train(False)      # chunked all-reduce of the whole gradient buffer
assert model._color_all is not None and dense._color_all is None
# reference: one process, gradients of the two views averaged by hand, plain optimizer.step
ref = fresh()
for step in range(3):
    ref.optimizer.lrs["means"] = ref.config.lr_means_at(step)     # the schedule train_step applies
    g = torch.zeros_like(ref.params.grad)
    for r in range(2):
        ref.forward_backward(views[(step * 2 + r) % 4])
        g += ref.params.grad
    ref.params.grad.copy_(g * 0.5)
    ref.optimizer.step()
torch.cuda.synchronize()
d = (ref.params.flat - model.params.flat).abs().max().item()
d2 = (ref.params.flat - dense.params.flat).abs().max().item()
assert d < 1e-6 and d2 < 1e-6, (d, d2)
dp.barrier()
# sync-free intersection budget under data parallelism: a forced overflow (capacity far too small) is
# agreed across the ranks on the device, noticed by both at the same step and replayed; the result is
# bit-identical to the synchronous budget and the replicas stay identical
def run_budget(speculative):
    m = fresh()
    if speculative:
        m.enable_speculative_budget(capacity=3000, max_in_flight=2)
    for step in range(7):
        m.train_step(views[dp.views_for_step(step, 4)], dp)
    m.flush()
    torch.cuda.synchronize()
    return m
spec, base = run_budget(True), run_budget(False)
assert spec.speculative_replays > 0 and spec.step == base.step == 7 and spec.optimizer.t == base.optimizer.t == 7
assert torch.equal(spec.params.flat, base.params.flat), (spec.params.flat - base.params.flat).abs().max().item()
assert torch.equal(spec.optimizer.exp_avg_sq, base.optimizer.exp_avg_sq)
dp.assert_replicas_identical(spec.params.flat)
dp.barrier()
# densification under data parallelism: statistics are reduced (sum, sum, max) and the split sampler
# is seeded from the step, so both replicas refine identically
from touch_gs_amd.densify import DensifyConfig
dm = fresh()
dm.enable_densification(DensifyConfig(warmup_length=2, refine_every=4, densify_grad_thresh=1e-5,
                                      densify_size_thresh=0.02, cull_alpha_thresh=0.01, reset_alpha_every=0))
for step in range(9):
    dm.train_step(views[dp.views_for_step(step, 4)], dp)
torch.cuda.synchronize()
n = torch.tensor([dm.params.N], dtype=torch.int64)
lo, hi = n.clone(), n.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert int(lo) == int(hi) and dm.params.N != N, (int(lo), int(hi), N)
dp.assert_replicas_identical(dm.params.flat)
dp.assert_replicas_identical(dm.optimizer.exp_avg_sq)
dp.barrier()
# ... and with the sync-free budget on top (refinements are its barrier points; a forced overflow is
# replayed, overflowed frames add nothing to the refinement statistics): same Gaussians, bit for bit
ds = fresh()
ds.enable_densification(DensifyConfig(warmup_length=2, refine_every=4, densify_grad_thresh=1e-5,
                                      densify_size_thresh=0.02, cull_alpha_thresh=0.01, reset_alpha_every=0))
ds.enable_speculative_budget(capacity=3000, max_in_flight=2)
for step in range(9):
    ds.train_step(views[dp.views_for_step(step, 4)], dp)
ds.flush()
torch.cuda.synchronize()
assert ds.speculative_replays > 0 and ds.step == dm.step == 9
assert ds.params.N == dm.params.N and torch.equal(ds.params.flat, dm.params.flat), (ds.params.N, dm.params.N)
assert torch.equal(ds.optimizer.exp_avg_sq, dm.optimizer.exp_avg_sq)
dp.assert_replicas_identical(ds.params.flat)
dp.barrier()
# ADVICE r3: the capacity fits ONE rank's frames but not the other's (rank 1 renders at half resolution: a quarter of
# the pairs).  Every step rank 0 overflows is voided on both ranks (agreed verdict) and replayed; rank 1, whose own
# frames fitted, must not count them twice in the refinement statistics (density.accumulate is guarded by the agreed
# verdict, not by the rank's own status word): bit-identical to the synchronous budget
from touch_gs_amd import ops
views2 = [views[v] if v % 2 == 0 else make_view(N, W // 2, H // 2, deg, 7, dev, view=v, n_views=4) for v in range(4)]
b0 = ops.IntersectBudget()
v0 = views2[dp.views_for_step(0, 4)]
p0 = fresh().params
ops.project_bin_sort(v0.cam, p0.means, p0.log_scales, p0.quats, p0.opac_logit, p0.sh, deg, b0)
mine = [None, None]
dist.all_gather_object(mine, (int(b0.last_n), int(b0.last_need)))
cap = (mine[1][1] + mine[0][0]) // 2
assert mine[1][1] < cap < mine[0][0], mine        # rank 1 certainly fits, rank 0 certainly does not
def run_asym(speculative):
    m = fresh()
    m.enable_densification(DensifyConfig(warmup_length=2, refine_every=4, densify_grad_thresh=1e-5,
                                         densify_size_thresh=0.02, cull_alpha_thresh=0.01, reset_alpha_every=0))
    if speculative:
        m.enable_speculative_budget(capacity=cap, max_in_flight=2)
    for step in range(9):
        m.train_step(views2[dp.views_for_step(step, 4)], dp)
    m.flush()
    torch.cuda.synchronize()
    return m
da, db = run_asym(True), run_asym(False)
assert da.speculative_replays > 0 and da.step == db.step == 9
assert da.params.N == db.params.N != N and torch.equal(da.params.flat, db.params.flat), (da.params.N, db.params.N, mine, cap)
dp.assert_replicas_identical(da.params.flat)
dp.barrier()
# front prefetch of the data-parallel form: with the rank's next view announced, the geometry Adam (last kernel of
# the step) also runs that view's K1 (tgs_adam_geom_project_next) and the next step only scans, fills and sorts --
# bit-identical to the unannounced sequence, also with densification statistics (radii) and the sync-free budget
# across a forced overflow
def run_announced(announce, speculative):
    m = fresh()
    m.enable_densification(DensifyConfig(warmup_length=1000, refine_every=1000))
    if speculative:
        m.enable_speculative_budget(capacity=3000, max_in_flight=2)
    taken = 0
    for step in range(7):
        nxt = views[dp.views_for_step(step + 1, 4)] if announce else None
        pre = getattr(m, "_prefetch_ready", None)
        taken += int(pre is not None and pre.front_issued)
        m.train_step(views[dp.views_for_step(step, 4)], dp, next_view=nxt)
    m.flush()
    torch.cuda.synchronize()
    return m, taken
for speculative in (False, True):
    (fa, taken), (fb, none) = run_announced(True, speculative), run_announced(False, speculative)
    assert taken >= 3 and none == 0, (taken, none)
    assert torch.equal(fa.params.flat, fb.params.flat), (fa.params.flat - fb.params.flat).abs().max().item()
    assert torch.equal(fa.optimizer.exp_avg, fb.optimizer.exp_avg) and torch.equal(fa.optimizer.exp_avg_sq, fb.optimizer.exp_avg_sq)
    assert torch.equal(fa.density.grad_norm_sum, fb.density.grad_norm_sum) and torch.equal(fa.density.max_radius, fb.density.max_radius)
    dp.assert_replicas_identical(fa.params.flat)
dp.barrier()
if dp.rank == 0: print("DP_OK", d)
'''


def test_data_parallel_train_step_two_ranks_one_gpu(dev, tmp_path):
    """Both DP steps -- factored (all-gather of colour gradients + all-reduce of geometry gradients,
    SH gradient rebuilt in the optimizer kernel) and dense (chunked all-reduce on a side stream +
    range-wise fused Adam) -- equal a single process that averages the two views' gradients;
    replicas stay identical."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), root],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_trainer_two_ranks_overflow_before_eval_boundary(dev, tmp_path, transport):
    """the trainer CLI with two ranks (sharing the GPU over gloo), the sync-free budget forced to
    overflow (TGS_SPEC_CAPACITY), refinement every 4 steps and eval / save boundaries every 5: every rank
    drains its pending verdicts at the same steps (a rank-0-only flush left the replayed collectives
    unmatched), so the run completes, refines, and ends with identical replicas (train.main asserts it)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGS_DIST_BACKEND="gloo", TGS_SPEC_CAPACITY="3000", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TGS_DP_TRANSPORT=transport)     # "ipc": the peer buffers are rebuilt at every refinement (N changes)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", free_port(), "-m", "touch_gs_amd.train",
                        "--synthetic", "4000", "160", "96", "--sh-degree", "1", "--max-num-iterations", "14",
                        "--steps-per-eval", "5", "--steps-per-save", "10", "--warmup-length", "4", "--refine-every", "4",
                        "--output-dir", str(out)],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "step 5:" in r.stdout and "step 10:" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("{") and "psnr" in l]
    assert line and json.loads(line[-1])["psnr"] > 0
    assert "speculative replays" in r.stdout and "refinements" in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("deg,interval,clamp", [(3, 0, False), (1, 0, False), (3, 1000, False), (3, 2, False),
                                                 (3, 0, True)])
def test_fused_backward_adam_equals_separate_kernels(dev, deg, interval, clamp):
    """tgs_project_bwd_adam == tgs_project_bwd followed by tgs_adam_step (same Adam arithmetic);
    interval > 0 ramps the ACTIVE degree below the stored one (1000: stays at 0; 2: 0,0,1 over the
    three steps), exercising the zero-gradient rows of the fused kernel."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H = 4100, 160, 96   # N not a multiple of 256: exercises the ragged last group
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(2)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    if clamp:   # alpha = 0.999 clamp regime (App. B.7 pass-through gradient): 1 in 5 at opacity logit 12
        P["opac_logit"][::5] = 12.0
        P["log_scales"][::5] += 1.5
    def run(fuse, steps):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=interval), params)
        m.fuse_adam = fuse
        assert m.optimizer.can_fuse_with_backward(m.active_sh_degree())
        for step in range(steps):
            m.train_step(views[step % 2])
        return m
    # one step: identical inputs, identical Adam arithmetic -> agreement to rounding
    a, b = run(True, 1), run(False, 1)
    for x, y, name in ((a.params.flat, b.params.flat, "params"), (a.optimizer.exp_avg, b.optimizer.exp_avg, "m"),
                       (a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq, "v")):
        scale = y.abs().max().item()
        assert (x - y).abs().max().item() < 1e-6 * scale + 1e-12, name
    # three steps: 1-ulp differences may flip a threshold decision for a few Gaussians -> statistical
    a, b = run(True, 3), run(False, 3)
    assert a.optimizer.t == b.optimizer.t == 3
    d = (a.params.flat - b.params.flat).abs()
    assert torch.quantile(d[::7].float(), 0.999).item() < 1e-5 * b.params.flat.abs().max().item()


def test_densification_clone_split_cull(dev):
    from touch_gs_amd.densify import DensifyConfig
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 4000, 160, 96, 1
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    P["opac_logit"][:200] = -6.0   # nearly transparent -> must be culled
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    model.enable_densification(DensifyConfig(warmup_length=4, refine_every=8, densify_grad_thresh=1e-5,
                                            densify_size_thresh=0.02, cull_alpha_thresh=0.01, reset_alpha_every=0))
    for step in range(8):
        model.train_step(views[step % 4])
    info = model.last_refine
    assert info["before"] == N and info["culled"] >= 200 and info["cloned"] + info["split"] > 0
    assert model.params.N == info["after"] and info["after"] != N
    assert model.optimizer.exp_avg.shape == model.params.flat.shape and model.optimizer.t == 8
    n1 = model.params.N
    # a checkpoint taken after refinement restores into a model built with the original count
    fresh_params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    resumed = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), fresh_params)
    resumed.load_state_dict(model.state_dict())
    assert resumed.params.N == model.params.N != N and torch.equal(resumed.params.flat, model.params.flat)
    assert torch.equal(resumed.optimizer.exp_avg_sq, model.optimizer.exp_avg_sq) and resumed.optimizer.t == model.optimizer.t
    for step in range(8, 12):   # training continues on the refined set (fused and unfused paths)
        model.train_step(views[step % 4])
    assert torch.isfinite(model.params.flat).all() and model.params.N == n1
    out = model.get_outputs(views[0].cam)
    assert out["rgb"].shape == (H, W, 3) and torch.isfinite(out["rgb"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("deg,stride", [(0, 1), (1, 4), (2, 16), (3, 16)])
def test_spherical_harmonics_op_matches_oracle(deg, stride):
    """tgs_sh_fwd / tgs_sh_bwd against the fp64 oracle basis; rows above the active degree get zero
    gradient; view directions are normalised in-kernel."""
    from oracle import torch_oracle as TO
    from touch_gs_amd import ops
    g = torch.Generator().manual_seed(5 + deg)
    N = 777
    dirs = torch.randn(N, 3, generator=g) * 3.0
    coeffs = torch.randn(N, stride, 3, generator=g)
    v = torch.randn(N, 3, generator=g)
    K = (deg + 1) ** 2
    Y = TO.sh_basis(deg, (dirs / dirs.norm(dim=-1, keepdim=True)).double())        # [N,K]
    want = torch.einsum("nk,nkc->nc", Y, coeffs[:, :K].double())
    want_v = torch.zeros(N, stride, 3, dtype=torch.float64)
    want_v[:, :K] = Y[:, :, None] * v.double()[:, None, :]

    c = coeffs.cuda().requires_grad_(True)
    got = ops.spherical_harmonics(deg, dirs.cuda(), c)
    got.backward(v.cuda())
    assert torch.allclose(got.detach().cpu().double(), want, rtol=1e-5, atol=1e-5)
    assert torch.allclose(c.grad.cpu().double(), want_v, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_resolution_schedule_equals_training_on_downscaled_views(dev):
    """Splatfacto's coarse-to-fine schedule (ModelConfig.num_downscales / resolution_schedule): the
    step at downscale factor d is bit for bit the step of a schedule-free model on ``view.downscaled(d)`` (camera
    intrinsics / d, sides floored, colour image bilinear, depth and uncertainty nearest so that 0 stays
    "unsupervised"), the factor follows 2^max(n - step // every, 0), under the speculative budget a replayed step is
    downscaled once, and evaluation renders stay at full resolution."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 4000, 200, 120, 1
    views = [make_view(N, W, H, deg, 17, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, deg, 18)

    def fresh(**kw):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0, **kw), params)

    v2 = views[0].downscaled(2)
    assert v2.rgb.shape == (60, 100, 3) and v2.depth.shape == (60, 100) and v2.cam.W == 100 and abs(v2.cam.fx - views[0].cam.fx / 2) < 1e-9
    assert v2 is views[0].downscaled(2) and views[0].downscaled(1) is views[0]
    assert set(torch.unique(v2.depth).tolist()) <= set(torch.unique(views[0].depth).tolist())   # nearest: no invented depths
    cfg = ModelConfig(num_downscales=2, resolution_schedule=3)
    assert [cfg.downscale_factor(s) for s in (0, 2, 3, 5, 6, 100)] == [4, 4, 2, 2, 1, 1]
    a, b = fresh(num_downscales=2, resolution_schedule=3), fresh()
    a.enable_speculative_budget(capacity=2000)        # too small: the first steps overflow and are replayed
    shapes = []
    for s in range(8):
        v = views[s % 4]
        a.train_step(v, next_view=views[(s + 1) % 4])
        b.train_step(v.downscaled(cfg.downscale_factor(s)))
        shapes.append(tuple(a.last["rgb"].shape[:2]))
    a.flush()
    assert getattr(a, "speculative_replays", 0) > 0
    assert shapes == [(30, 50)] * 3 + [(60, 100)] * 3 + [(120, 200)] * 2, shapes
    assert torch.equal(a.params.flat, b.params.flat)
    assert a.get_outputs(views[0].cam)["rgb"].shape == (H, W, 3)


@pytest.mark.parametrize("deg,interval", [(3, 0), (1, 0), (3, 1000)])
def test_gathered_sh_adam_equals_dense_adam_single_rank(dev, deg, interval):
    """world = 1: tgs_project_bwd_color + tgs_adam_step_sh_gathered + geometry tgs_adam_step equals
    tgs_project_bwd + tgs_adam_step (same products, same Adam arithmetic);
    interval=1000 keeps the active degree at 0 with degree-3 storage (zero-gradient rows)."""
    from touch_gs_amd import parallel
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H = 4100, 160, 96
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(2)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=interval), params)
    a, b = fresh(), fresh()
    dp = parallel.GradSync(0, 1, 0)
    block = torch.zeros(3 * N + 4, device=dev)
    allc = torch.zeros(1, 3 * N + 4, device=dev)
    for step in range(2):
        view = views[step]
        adeg = a.active_sh_degree()
        a.forward_backward(view, color_block=block)
        dp.gather_color_reduce_geom_and_step(a.params.grad[:a.optimizer.geom_end()], block, allc,
                                             lambda c, sc: a.optimizer.step_sh_gathered(1, adeg, c, sc),
                                             a.optimizer.step_range, a.optimizer.begin_step)
        a.step += 1
        b.forward_backward(view)
        b.optimizer.step()
        b.step += 1
    assert (block[3 * N:3 * N + 3].cpu() - torch.tensor(views[1].cam.position())).abs().max() < 1e-5
    # same products and the same Adam arithmetic, but evaluated in differently specialised kernels:
    # FMA contraction may differ by an ulp at degree >= 2
    for x, y, name in ((a.params.flat, b.params.flat, "params"), (a.optimizer.exp_avg, b.optimizer.exp_avg, "m"),
                       (a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq, "v")):
        scale = y.abs().max().item()
        assert (x - y).abs().max().item() <= 2e-6 * scale + 1e-12, name


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [4, 3, 16])
def test_pipelined_color_exchange_is_bit_identical_to_the_unchunked_form(dev, chunks):
    """Row-chunked K8 (tgs_project_bwd_color_rows) + per-chunk SH Adam (tgs_adam_step_sh_gathered_rows) --
    the pipelined exchange of the data-parallel step -- against the one-piece calls: parameters, both moment
    buffers, geometry gradients and v_xy bit for bit (world = 1; N not a multiple of the chunk grain)."""
    from touch_gs_amd import parallel
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 4100, 160, 96, 3
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(2)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    a, b = fresh(), fresh()
    dp = parallel.GradSync(0, 1, 0, color_chunks=chunks)
    rows = dp.color_chunk_rows(N)
    assert 1 < len(rows) <= min(chunks, N // 256) and rows[0][0] == 0 and rows[-1][1] == N
    assert all(r[0] % 256 == 0 for r in rows) and all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    block = torch.zeros(3 * N + 4, device=dev)
    allc = torch.zeros(1, 3 * N + 4, device=dev)
    blocks = [torch.zeros(3 * (e - s) + 4, device=dev) for s, e in rows]
    blocks_all = [torch.zeros(1, 3 * (e - s) + 4, device=dev) for s, e in rows]
    for step in range(2):
        view = views[step]
        a.forward_backward(view, color_block=block, want_v_xy=True)
        ga = a.params.grad[:a.optimizer.geom_end()].clone()
        dp.gather_color_reduce_geom_and_step(a.params.grad[:a.optimizer.geom_end()], block, allc,
                                             lambda c, sc: a.optimizer.step_sh_gathered(1, deg, c, sc),
                                             a.optimizer.step_range, a.optimizer.begin_step)
        b.forward_backward(view, color_block=(rows, blocks), want_v_xy=True)
        dp.pipelined_color_exchange_and_step(
            b.params.grad[:b.optimizer.geom_end()], blocks, blocks_all, b._backward_chunk,
            lambda c, allb, sc: b.optimizer.step_sh_gathered(1, deg, allb, sc, rows=rows[c]),
            b.optimizer.step_range, b.optimizer.begin_step)
        assert torch.equal(ga, b.params.grad[:b.optimizer.geom_end()])
        assert torch.equal(a.last["v_xy"], b.last["v_xy"])
        assert torch.equal(torch.cat([blk[:-4] for blk in blocks]), block[:3 * N])
        for blk in blocks:
            assert torch.equal(blk[-4:], block[3 * N:])       # every chunk carries the camera trailer
        a.step += 1
        b.step += 1
    assert torch.equal(a.params.flat, b.params.flat)
    assert torch.equal(a.optimizer.exp_avg, b.optimizer.exp_avg)
    assert torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)


@pytest.mark.gpu
def test_step_graph_replay_equals_eager_steps(dev):
    """hipGraph replay of the fused train step (one captured graph per view, Adam bias corrections
    read from device memory) is bit-identical to launching the kernels one by one."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 6000, 160, 96, 3
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(3)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    eager, graphed = fresh(), fresh()
    graphed.capture_step_graphs(views)
    assert len(graphed._graphs) == 3
    for step in range(7):
        v = views[step % 3]
        eager.train_step(v)
        graphed.train_step(v)
    torch.cuda.synchronize()
    assert graphed.optimizer.t == eager.optimizer.t == 7 and graphed.step == 7
    assert graphed.budget.check() > 0
    for x, y in ((graphed.params.flat, eager.params.flat), (graphed.optimizer.exp_avg, eager.optimizer.exp_avg),
                 (graphed.optimizer.exp_avg_sq, eager.optimizer.exp_avg_sq)):
        d = (x - y).abs()
        assert torch.equal(x, y), (d.max().item(), int((d > 0).sum()), torch.get_default_dtype())
    assert torch.equal(graphed.last["rgb"], eager.last["rgb"])
    # a view without a graph falls back to the eager path
    other = make_view(N, W, H, deg, 7, dev, view=3, n_views=4)
    graphed.train_step(other); eager.train_step(other)
    assert torch.equal(graphed.params.flat, eager.params.flat)


def _write_reference_scene(root, dev, N, W, H, deg, n_views=6, seed=11):
    """A scene directory in the format the reference's plumbing leaves behind:
    transforms.json with per-frame file_path / OpenGL camera->world transform_matrix / depth_file_path /
    uncertainty_file_path (utils/add_depth_file_path_to_transforms.py:37-50), 8-bit RGB PNGs, 16-bit
    millimetre depth + uncertainty PNGs (utils/fuse_touch_vision.py:372-376), touch seed points
    (utils/create_point_cloud_from_touches.py:243-244).  Returns the camera centres."""
    import json, os
    from PIL import Image
    from touch_gs_amd import plumbing
    from touch_gs_amd.scene import make_view
    for d in ("images", "fused_output_dir", "fused_output_dir_uncertainty"):
        os.makedirs(root / d)
    frames, centres = [], []
    for i in range(n_views):
        v = make_view(N, W, H, deg, seed, dev, view=i, n_views=n_views)
        Image.fromarray((v.rgb.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()).save(root / "images" / f"{i}.png")
        plumbing.write_png16(str(root / "fused_output_dir" / f"{i}.png"), plumbing.to_uint16_mm(v.depth.cpu().numpy().astype(np.float64)))
        plumbing.write_png16(str(root / "fused_output_dir_uncertainty" / f"{i}.png"),
                             plumbing.to_uint16_mm(v.uncertainty.cpu().numpy().astype(np.float64)))
        c2w = np.linalg.inv(np.asarray(v.cam.viewmat, dtype=np.float64)) @ np.diag([1.0, -1.0, -1.0, 1.0])
        centres.append(c2w[:3, 3])
        frames.append({"file_path": f"images/{i}.png", "transform_matrix": c2w.tolist()})
    meta = {"fl_x": v.cam.fx, "fl_y": v.cam.fy, "cx": v.cam.cx, "cy": v.cam.cy, "w": W, "h": H, "frames": frames}
    plumbing.add_depth_file_paths(meta, "fused_output_dir", "fused_output_dir_uncertainty")
    (root / "transforms.json").write_text(json.dumps(meta))
    rng = np.random.default_rng(0)
    np.save(root / "points_touch.npy", rng.uniform(-0.5, 0.5, (200, 3)) + np.array([0, 0, 4.0]))
    np.save(root / "points_colors.npy", rng.uniform(0, 255, (200, 3)))
    return centres


def test_trainer_on_disk_dataset_in_reference_format(dev, tmp_path):
    """End to end on the on-disk contract the reference's plumbing produces:
    transforms.json (fl_x.., per-frame OpenGL camera->world, depth_file_path, uncertainty_file_path),
    8-bit RGB PNGs, 16-bit millimetre depth / uncertainty PNGs, touch seed points -> Scene loader ->
    trainer -> eval.json + render dump."""
    import json, os
    from PIL import Image
    from touch_gs_amd import plumbing, train
    from touch_gs_amd.dataset import Scene
    from touch_gs_amd.scene import make_view
    N, W, H, deg = 3000, 96, 64, 1
    root = tmp_path / "scene"
    centres = _write_reference_scene(root, dev, N, W, H, deg)

    sc = Scene(str(root), train_split_fraction=0.8, device=dev)
    assert len(sc.views) == 6 and len(sc.i_train) == 5 and len(sc.i_eval) == 1
    # poses are centred and scaled by 1 / max|t|, depths by the same factor (legacy/dataparser_tactile.py:222-235)
    c = np.stack(centres); c = c - c.mean(0)
    assert abs(sc.scale - 1.0 / np.abs(c).max()) < 1e-9
    v0 = make_view(N, W, H, deg, 11, dev, view=0, n_views=6)
    got, want = sc.views[0].depth, v0.depth * sc.scale
    assert torch.allclose(got, want, atol=1e-3 * sc.scale + 1e-6)            # millimetre quantisation
    assert sc.seed_points()[0].shape == (200, 3)

    run = train.main(["--data", str(root), "--train-split-fraction", "0.8", "--max-num-iterations", "10",
                      "--steps-per-eval", "5", "--steps-per-save", "10", "--sh-degree", "1", "--num-gaussians", "2000",
                      "--depth-loss-mult", "0.005", "--uncertainty-weight", "0.01", "--output-dir", str(tmp_path / "out"),
                      "--render-output", str(tmp_path / "renders")])
    ev = json.load(open(os.path.join(run, "eval.json")))
    assert {"psnr", "ssim", "depth_mse", "supervised_depth_mse"} <= set(ev["results"])
    assert len(os.listdir(tmp_path / "renders" / "rgb")) == 1 and len(os.listdir(tmp_path / "renders" / "depth")) == 1
    # uncertainty units (dataset.py docstring, UNVERIFIED-PRIOR): default "linear" = the depth image's factor;
    # "variance" = its square; "none" = as stored.  The choice is recorded in the run's config.json.
    assert sc.uncertainty_scaling == "linear"
    assert torch.allclose(sc.views[0].uncertainty, v0.uncertainty * sc.scale, atol=1e-3 * sc.scale + 1e-7)
    sv = Scene(str(root), train_split_fraction=0.8, device=dev, uncertainty_scaling="variance")
    assert torch.allclose(sv.views[0].uncertainty, v0.uncertainty * sc.scale ** 2, atol=1e-3 * sc.scale ** 2 + 1e-7)
    sn = Scene(str(root), train_split_fraction=0.8, device=dev, uncertainty_scaling="none")
    assert torch.allclose(sn.views[0].uncertainty, v0.uncertainty, atol=1e-3 + 1e-7)
    rec = json.load(open(os.path.join(run, "config.json")))["scene"]
    assert rec["uncertainty_scaling"] == "linear" and abs(rec["uncertainty_factor"] - sc.scale) < 1e-12

    # the reference's run_eval step (experiment_utils/run_eval.py:37-57) with IS_REAL_WORLD exported
    # (scripts/train_bunny_real.sh:54): <exp>/<exp>_<k>.json newest first, gt_* keys, render dump
    from touch_gs_amd import run_eval
    for d in ("realsense_depths", "touch_depth"):
        os.makedirs(root / d)
    for i in range(6):
        v = make_view(N, W, H, deg, 11, dev, view=i, n_views=6)
        d_m = v.depth.cpu().numpy().astype(np.float64)
        plumbing.write_png16(str(root / "realsense_depths" / f"{i}.png"), plumbing.to_uint16_mm(d_m))
        obj = np.zeros_like(d_m); obj[16:48, 24:72] = d_m[16:48, 24:72]
        plumbing.write_png16(str(root / "touch_depth" / f"{i}.png"), plumbing.to_uint16_mm(obj))
    old = os.environ.get("IS_REAL_WORLD")
    os.environ["IS_REAL_WORLD"] = "True"
    cwd = os.getcwd()
    try:
        os.chdir(tmp_path)
        done = run_eval.main(["--input_dir", os.path.dirname(run), "--output_dir", str(tmp_path / "experiments"),
                              "--exp_name", "disk_exp", "--past_n_trials", "1"])
    finally:
        os.chdir(cwd)
        if old is None:
            os.environ.pop("IS_REAL_WORLD")
        else:
            os.environ["IS_REAL_WORLD"] = old
    assert [os.path.basename(p) for p in done] == ["disk_exp_1.json"]
    res = json.load(open(done[0]))["results"]
    assert {"psnr", "ssim", "lpips", "depth_mse", "supervised_depth_mse", "gt_depth_mse", "gt_object_depth_mse"} <= set(res)
    assert abs(res["psnr"] - ev["results"]["psnr"]) < 1e-3 and res["gt_depth_mse"] > 0
    assert len(os.listdir(tmp_path / "disk_exp_renders" / "rgb")) == 1


@pytest.mark.parametrize("announce", [False, True])
@pytest.mark.parametrize("capacity", [0, 3000])
def test_speculative_budget_equals_synchronous_budget(dev, capacity, announce):
    """Sync-free training: status words are read late; capacity=3000 is far too small, so frames
    overflow, the sticky device word turns the following steps into no-ops, and the host replays them
    with grown buffers -- parameters, moments and step counters end bit-identical to the synchronous
    budget (capacity=0 = generous default: no overflow at all)."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 6000, 160, 96, 3
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=4), params)
    ref, spec = fresh(), fresh()
    spec.enable_speculative_budget(capacity=capacity, max_in_flight=3)
    for step in range(11):
        ref.train_step(views[step % 4])
        # the next view is announced: colour + front prefetch run ahead of the verdicts (the overflowing steps void
        # the optimizer kernel that would have produced them; the replay drops what a voided step announced)
        spec.train_step(views[step % 4], next_view=views[(step + 1) % 4] if announce else None)
    spec.flush()
    torch.cuda.synchronize()
    assert spec.step == ref.step == 11 and spec.optimizer.t == ref.optimizer.t == 11
    replays = getattr(spec, "speculative_replays", 0)
    assert (replays > 0) == (capacity > 0)
    assert spec.budget.capacity >= int(spec.last["status"][0])
    for x, y in ((spec.params.flat, ref.params.flat), (spec.optimizer.exp_avg, ref.optimizer.exp_avg),
                 (spec.optimizer.exp_avg_sq, ref.optimizer.exp_avg_sq)):
        assert torch.equal(x, y)
    assert torch.equal(spec.last["rgb"], ref.last["rgb"])


def test_peer_wait_timeout_poisons_the_step_and_alloc_reports_its_memory_kind(dev):
    """ADVICE r4 / VERDICT r4 next #5c.  tgs_peer_alloc returns the kind of device memory it obtained and refuses what the
    caller did not allow (no silent downgrade to plain memory).  A tgs_peer_wait whose flag never arrives gives up after
    its timeout, records which flag was missing, and raises the two poison words -- with the sticky word raised
    tgs_dp_agree_overflow declares the step void whatever the ranks' flags say, and a guarded optimizer kernel behind it
    changes nothing."""
    import ctypes as C
    from touch_gs_amd import _lib, ops
    from touch_gs_amd.optim import FusedAdam, GaussianParams
    lib = _lib.load()
    base, handle, kind = C.c_void_p(), (C.c_ubyte * 64)(), C.c_int(0)
    _lib.check(lib.tgs_peer_alloc(4096, 1 | 2, C.byref(base), handle, C.byref(kind)), "tgs_peer_alloc")
    assert kind.value in (1, 2)                                  # uncached or fine-grained, as asked
    try:
        assert lib.tgs_peer_alloc(4096, 0, C.byref(C.c_void_p()), handle, None) == -1        # nothing allowed: refused
        sticky = torch.zeros(1, dtype=torch.int32, device=dev)
        verdict = torch.zeros(2, dtype=torch.int32, device=dev)
        flags = (C.c_void_p * 1)(base.value)                     # a flag word (zeroed) that nobody will raise
        err = base.value + 64
        t0 = time.perf_counter()
        _lib.check(lib.tgs_peer_wait(1, flags, 5, err, C.c_float(0.05), _lib.ptr(sticky), _lib.ptr(verdict) + 4, None), "tgs_peer_wait")
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 5.0                    # gave up after ~50 ms, not after the default 20 s
        # the error word lives in the peer buffer: read it through a device tensor view, as PeerExchange does
        from touch_gs_amd.parallel import _RawDeviceArray
        e = torch.as_tensor(_RawDeviceArray(err, 1, "<i4"), device=dev)
        assert int(e.item()) == 1                                # 1 + index of the missing flag
        assert int(sticky.item()) == 1 and verdict.tolist() == [0, 1]
        # the agreed verdict with the sticky word raised: void, whatever the gathered flags say
        N = 64
        blocks = torch.zeros(2, 3 * N + 4, device=dev)           # two ranks' colour blocks, overflow pads = 0
        out = torch.zeros(2, dtype=torch.int32, device=dev)
        ops.dp_agree_overflow(2, N, blocks, out, sticky)
        assert out.tolist() == [0, 1]
        sticky.zero_()
        ops.dp_agree_overflow(2, N, blocks, out, sticky)
        assert out.tolist() == [0, 0]
        # a guarded optimizer kernel behind the poisoned verdict is a no-op
        gp = GaussianParams.allocate(N, 16, dev)
        gp.flat.copy_(torch.randn_like(gp.flat)); gp.grad.copy_(torch.randn_like(gp.flat))
        opt = FusedAdam(gp, dict(means=1e-3, log_scales=1e-3, quats=1e-3, opac_logit=1e-3, sh_dc=1e-3, sh_rest=1e-3))
        before = gp.flat.clone()
        opt.step(guard=verdict)
        torch.cuda.synchronize()
        assert torch.equal(gp.flat, before)
    finally:
        lib.tgs_peer_free(base)


def test_learned_list_hint_skips_launches_and_a_broken_hint_is_replayed(dev):
    """Speculative budget: after LIST_HINT_AFTER settled frames the trainer bounds the longest tile list by 1.5 x the
    longest it has seen and the sort launches for longer list classes are no longer issued.  Here the first views have
    lists of a few dozen entries and a later view looks along a row of Gaussians (one tile list beyond 1024): that
    frame breaks the learned bound, is voided on the device, and the replay (hint off) puts the model exactly where
    the trainer without the hint ends -- parameters and moments bit for bit."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 2300, 160, 96, 3      # 1000 scattered Gaussians (lists of a few dozen entries) + the row
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=8) for v in (0, 1, 2, 7)]   # (7 is opposite to none of the others)
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    # 1300 small Gaussians on the optical axis of view 3 only: one tile of that view gets a list > 1024
    c2w = torch.linalg.inv(torch.tensor(views[3].cam.viewmat, dtype=torch.float64).reshape(4, 4))
    g = torch.Generator().manual_seed(5)
    z = 1.0 + 8.0 * torch.rand(1300, generator=g, dtype=torch.float64)     # a long row: from the side it crosses many tiles
    pts = torch.stack([0.002 * torch.randn(1300, generator=g, dtype=torch.float64) * z,
                       0.002 * torch.randn(1300, generator=g, dtype=torch.float64) * z, z, torch.ones(1300, dtype=torch.float64)], 1)
    P["means"][:1300] = (pts @ c2w.T)[:, :3].to(P["means"].dtype)
    P["log_scales"][:1300] = -5.5
    P["opac_logit"][:1300] = -3.0          # faint: the pixels behind them stay live, the whole list is walked

    def run(hint):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
        m.list_hint, m.LIST_HINT_AFTER = hint, 4
        m.enable_speculative_budget(capacity=0, max_in_flight=2)
        order = [0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 3, 1, 2, 3, 0]
        hints = []
        for i, v in enumerate(order):
            m.train_step(views[v], next_view=views[order[i + 1]] if i + 1 < len(order) else None)
            hints.append(m.budget.max_list_hint)
        m.flush()
        torch.cuda.synchronize()
        return m, hints

    a, ha = run(True)
    b, hb = run(False)
    assert all(h == -1 for h in hb) and getattr(b, "speculative_replays", 0) == 0
    assert 0 < max(ha[:10]) <= 1024, ha          # a bound below the wg4 class was learned ...
    assert getattr(a, "speculative_replays", 0) > 0, ha          # ... view 3 broke it and was replayed
    assert a._seen_longest > 1024
    assert a.step == b.step == 15 and a.optimizer.t == b.optimizer.t == 15
    for x, y in ((a.params.flat, b.params.flat), (a.optimizer.exp_avg, b.optimizer.exp_avg),
                 (a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)):
        assert torch.equal(x, y)


_RCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dp = parallel.init_from_env()                      # backend "nccl" == RCCL on ROCm
assert dist.is_initialized() and dist.get_backend() == "nccl" and dp.world == 1 and dp.active
dev = torch.device("cuda:0")
N, W, H, deg = 4001, 160, 96, 3
views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(2)]
P, _ = synthetic_gaussians(N, W, H, deg, 99)
def fresh():
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
dp.n_chunks = 3
dp.color_chunks = 3          # pipelined factored exchange: K8 / gather / SH Adam in 3 row chunks
def train(factored, sync):
    m = fresh()
    m.dp_factored_sh = factored
    m.fuse_adam = False
    for step in range(3):
        m.train_step(views[step % 2], sync)
    torch.cuda.synchronize()
    return m
ref = train(True, None)                             # single process: K8 -> flat gradient -> Adam, no collectives
a = train(True, dp)                                 # all_gather_into_tensor + all_reduce over RCCL
b = train(False, dp)                                # chunked all_reduce of the flat buffer over RCCL
assert a._color_all is not None and dp.bytes_per_step > 0
dp.timing = True
a.train_step(views[1], dp)
rep = dp.comm_report()
assert rep and rep["all_gather_ms"] > 0 and rep["all_reduce_ms"] > 0, rep
ea = (ref.params.flat - train(True, dp).params.flat).abs()
da = ea.max().item()
db = (ref.params.flat - b.params.flat).abs().max().item()
# dense exchange: the same kernels as the reference -> identical.  Factored exchange: the SH gradient is
# rebuilt by another kernel (and the geometry gradients come from the colour-mode instantiation of K8), so
# single gradients differ in the last bit -- which Adam's normalisation turns into a visible (but bounded
# by the learning rate) difference on the few elements whose gradient is ~0
assert db == 0.0 and da < 2e-5 and (ea > 1e-6).float().mean().item() < 1e-3, (da, db, (ea > 1e-6).float().mean().item())
dp.assert_replicas_identical(a.params.flat)
assert abs(dp.max_over_ranks(1.5) - 1.5) < 1e-12
dp.barrier()
dist.destroy_process_group()
print("RCCL_OK", da, db, rep)
'''


def test_rccl_single_rank_process_group(dev, tmp_path):
    """the RCCL (backend "nccl") code path of both GradSync forms -- all_gather_into_tensor
    + all_reduce on the side stream, and the chunked all_reduce -- executed on ONE GPU with a 1-rank
    process group (TGS_DP_FORCE_COLLECTIVES=1 issues the collectives even though world == 1); the
    result equals the single-process fused step."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               TGS_DP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TGS_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_FUSED_TAIL_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dp = parallel.init_from_env()
dev = torch.device("cuda", dp.local_rank)
N, W, H, deg = 20_000, 320, 208, 3
P, _ = synthetic_gaussians(N, W, H, deg, 17)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
m.spatial_sort()
views = [make_view(N, W, H, deg, 17, dev, view=v, n_views=4) for v in range(4)]
for v in views: v.valid_count()
for i in range(6):
    m.train_step(views[dp.views_for_step(i, 4)], dp, next_view=views[dp.views_for_step(i + 1, 4)])
torch.cuda.synchronize()
dp.assert_replicas_identical(m.params.flat)
if dp.rank == 0:
    torch.save(dict(flat=m.params.flat.cpu(), m=m.optimizer.exp_avg.cpu(), v=m.optimizer.exp_avg_sq.cpu(),
                    splats=m._prefetch_ready.front.splats.cpu(), issued=bool(m._prefetch_ready.front_issued)), sys.argv[2])
dp.barrier()
import torch.distributed as dist
dist.destroy_process_group()
print("TAIL_OK")
'''


@pytest.mark.parametrize("world", [1, 2])
def test_fused_data_parallel_tail_is_bit_identical_to_the_chunked_one(dev, tmp_path, world):
    """tgs_adam_sh_gathered_geom_project_next (round 6): the tail of a data-parallel step -- SH Adam of every row chunk from
    the gathered colour blocks, geometry Adam from the all-reduced gradients, the next view's colours and K1 -- as ONE
    launch (TGS_DP_FUSED_TAIL=1; the default only for one-rank groups) against the chunk-by-chunk sequence (=0), three row
    chunks, six steps with the next view announced: parameters, both Adam moments and the prefetched records of the next
    frame bit for bit.  world 1: a one-rank RCCL group (TGS_DP_FORCE_COLLECTIVES); world 2: two ranks share the GPU over gloo."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tail_worker.py"
    script.write_text(_FUSED_TAIL_WORKER)
    out = {}
    for mode in ("0", "1"):
        f = tmp_path / f"tail_{mode}.pt"
        env = dict(os.environ, TGS_DP_FUSED_TAIL=mode, TGS_DP_COLOR_CHUNKS="3", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                       TGS_DP_FORCE_COLLECTIVES="1")
            env.pop("TGS_DIST_BACKEND", None)
            cmd = [sys.executable, str(script), root, str(f)]
        else:
            env["TGS_DIST_BACKEND"] = "gloo"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                   "127.0.0.1", "--master-port", free_port(), str(script), root, str(f)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "TAIL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        out[mode] = torch.load(f)
    a, b = out["0"], out["1"]
    assert a["issued"] and b["issued"]
    for k in ("flat", "m", "v", "splats"):
        assert torch.equal(a[k], b[k]), k


_RESORT_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import parallel
from touch_gs_amd.densify import DensifyConfig
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dp = parallel.init_from_env()
dev = torch.device("cuda", dp.local_rank)
N, W, H, deg = 12_000, 320, 208, 3
P, _ = synthetic_gaussians(N, W, H, deg, 23)
P["log_scales"][::40] += 2.5          # a few hundred Gaussians that cover > 32 tiles: the re-sort deals them over the groups
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0, spatial_sort=True, resort_every_refines=1), params)
m.spatial_sort()
m.enable_densification(DensifyConfig(warmup_length=4, refine_every=4, reset_alpha_every=0, densify_grad_thresh=1e-7))
views = [make_view(N, W, H, deg, 23, dev, view=v, n_views=6) for v in range(6)]
for v in views: v.valid_count()
sorts = []
orig = m.spatial_sort
def counted():
    perm = orig()
    sorts.append(int((perm != torch.sort(perm).values).sum()))
    return perm
m.spatial_sort = counted
for i in range(13):
    m.train_step(views[dp.views_for_step(i, 6)], dp, next_view=views[dp.views_for_step(i + 1, 6)])
torch.cuda.synchronize()
dp.assert_replicas_identical(m.params.flat)
assert len(sorts) >= 2 and len(m._recent_cams) >= 2, (sorts, len(m._recent_cams))
dp.barrier()
import torch.distributed as dist
dist.destroy_process_group()
print("RESORT_OK", m.params.N, sorts)
'''


def test_balanced_resort_keeps_the_replicas_identical(dev, tmp_path):
    """The re-sort deals the long-run Gaussians over the binning groups from the tile counts of the cameras of the last
    steps (optim.balanced_order) -- and under data parallelism every rank has seen different cameras: the counts are summed
    over the ranks, so every replica computes the same permutation.  Two ranks share the GPU over gloo, refinement every 4
    steps with a full re-sort after each: the replicas stay bit-identical (they would not with per-rank counts)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "resort_worker.py"
    script.write_text(_RESORT_WORKER)
    env = dict(os.environ, TGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", free_port(), str(script), root], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RESORT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_two_gpus_rccl(dev, tmp_path):
    """First multi-GPU evidence wherever >= 2 GPUs are visible: bench.py --gpus 2 through
    torch.distributed.run over RCCL; the JSON line carries the exchange's bus bandwidth and the
    replicas stay identical (bench.py asserts it)."""
    import json, os, subprocess, sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", free_port(), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2", "--config", "cfg2"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["dp_exchange"]["all_gather_busbw_GBs"] > 0 and line["dp_exchange"]["transport"] == transport
    sweep = line["dp_exchange"]["busbw_sweep"]          # message-size sweep of the exchange primitives (token sizes on gloo)
    assert set(sweep) == {"12MB", "44MB"} and all(v["all_gather"]["busbw_GBs"] > 0 for v in sweep.values())
    assert ("peer_push" in sweep["12MB"]) == (transport == "ipc") and line["dp_exchange"]["replicas_identical"] is True


def test_bench_two_ranks_share_one_gpu(dev):
    """bench.py's N > 1 code path (rank rendezvous, per-rank views, barrier + max-over-ranks timing,
    exchange report, replica check) on a ONE-GPU box: two ranks share cuda:0 and exchange over gloo
    (TGS_DIST_BACKEND; RCCL refuses duplicate devices).  The driver launches the RCCL form of exactly
    this command on the 8-GPU node."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", free_port(), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", "cfg2", "--no-cpu-baseline", "--dp-report"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{"), r.stdout[:2000]   # ONE JSON line and nothing else on stdout
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["dp_exchange"]["replicas_identical"] is True
    assert line["config"]["parallelism"] == "dp2"
    # --dp-report: the same loop over every transport, each with its replica check (the collective one runs over gloo here)
    tr = line["dp_exchange"]["transports"]
    assert set(tr) == {"rccl", "ipc_flags_behind_kernel_boundary", "ipc_flags_in_kernel"}, tr
    for name, t in tr.items():
        assert t.get("replicas_identical") is True and t["ms_per_step"] > 0, (name, t)
    assert tr["ipc_flags_in_kernel"]["memory_kind"] in ("uncached", "fine-grained")


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
def test_bench_bare_command_launches_its_own_ranks(dev, transport):
    """`python bench.py --gpus 2` exactly as the driver invokes it (no torchrun, no rank environment):
    bench.py starts its own two ranks (here sharing cuda:0 over gloo) and prints ONE JSON line with
    n_gpus = 2, the exchange report and the replica check."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TGS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TGS_DP_TRANSPORT=transport)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--config", "cfg2"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # nothing but the JSON line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["dp_exchange"]["replicas_identical"] is True
    assert line["dp_exchange"]["all_gather_busbw_GBs"] > 0


def test_nerfstudio_adapter_core_trains(dev):
    """The nerfstudio adapter's core (everything DepthGSNerfstudioModel delegates to): six Splatfacto
    parameter groups as nn.Parameters, differentiable render, loss from a nerfstudio-shaped batch
    ([H,W,1] depth / uncertainty); a nerfstudio-style iteration -- loss.backward() + per-group Adam --
    reduces the loss, and the loss equals the core model's on the same view."""
    import types
    from touch_gs_amd.nerfstudio_plugin import PARAM_GROUP_LRS, AutogradGaussians
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 3000, 128, 80, 2
    view = make_view(N, W, H, deg, 5, dev)
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    cfg = types.SimpleNamespace(sh_degree=deg, ssim_lambda=0.2, depth_loss_mult=0.2,
                                depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS", uncertainty_weight=1.0)
    ag = AutogradGaussians(cfg, P["means"], torch.rand(N, 3), device=dev)
    groups = ag.param_groups()
    assert set(groups) == set(PARAM_GROUP_LRS) and all(isinstance(v[0], torch.nn.Parameter) for v in groups.values())
    opts = {k: torch.optim.Adam(v, lr=PARAM_GROUP_LRS[k] * 20, eps=1e-15) for k, v in groups.items()}
    batch = {"image": view.rgb, "depth_image": view.depth[..., None], "uncertainty": view.uncertainty[..., None]}
    losses = []
    for it in range(12):
        out = ag.render(view.cam)
        ld = ag.loss_dict(out, batch)
        assert set(ld) == {"main_loss", "depth_loss"} and ld["depth_loss"].ndim == 0
        loss = sum(ld.values())
        for o in opts.values():
            o.zero_grad()
        loss.backward()
        assert all(p[0].grad is not None and torch.isfinite(p[0].grad).all() for p in groups.values())
        for o in opts.values():
            o.step()
        losses.append(float(loss))
    assert losses[-1] < 0.9 * losses[0], losses
    m = ag.metrics_dict(ag.render(view.cam), batch)
    assert "psnr" in m and "depth_mse" in m


@pytest.mark.parametrize("W,H", [(160, 96), (208, 144), (1920, 1080)])
def test_pipelined_ssim_and_k7_bands_equal_the_sequential_step(dev, W, H, request):
    """SSIM pipelined by image bands behind K7 on a second stream (ModelConfig.pipeline_ssim;
    tgs_ssim_fwd_bwd_rows + tgs_rasterize_bwd_band): the bands partition the tiles, the band-wise SSIM
    gradient equals the whole-image one bit for bit, and three train steps leave the same parameters and
    Adam moments as the sequential form (whole-image SSIM, then one K7 launch)."""
    from touch_gs_amd import ops
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    # a band launch of K7 keeps one wave per tile: the whole-image launch it is compared with bit for bit must too
    # (in a chain-bound frame it would hand its long tiles to the four-wave kernel, tgs_set_k7_quad)
    request.addfinalizer(lambda before=ops.set_k7_quad(): ops.set_k7_quad(*before))
    ops.set_k7_quad(0)
    N = 4100 if W < 1000 else 60000
    views = [make_view(N, W, H, 3, 7, dev, view=v, n_views=4) for v in range(2)]
    P, _ = synthetic_gaussians(N, W, H, 3, 99)
    cam = views[0].cam
    # -- the band structure
    bands = ops.band_rows(cam)
    TW, TH = cam.tiles
    assert 1 <= len(bands) <= max(4, -(-TW * TH // 8192))
    assert bands[0][3] == 0 and bands[-1][4] == H and all(a[4] == b[3] for a, b in zip(bands, bands[1:]))
    assert all(y0 <= c0 <= c1 <= y1 for _, y0, y1, c0, c1 in bands)
    # -- band-wise SSIM == whole-image SSIM
    g = torch.Generator().manual_seed(1)
    img = torch.rand(H, W, 3, generator=g).to(dev)
    wgt = -0.2 / (3 * H * W)
    tot, v_full = ops.ssim_fwd_bwd(img, views[0].rgb, weight=wgt, reduce=False)
    lib = ops._lib.load()
    v_band = torch.full_like(v_full, float("nan"))
    scratch = torch.empty(9 * H * W, device=dev)
    n_p = ((W + 63) // 64) * (H // 12 + 2)
    sums = []
    for b, y0, y1, c0, c1 in bands:
        bp = torch.empty(n_p, device=dev)
        ops.check(lib.tgs_ssim_fwd_bwd_rows(W, H, ops.ptr(img), ops.ptr(views[0].rgb), ops.C.c_float(wgt), ops.ptr(bp), n_p,
                                            ops.ptr(v_band), ops.ptr(scratch), y0, y1, c0, c1, ops._stream()), "rows")
        sums.append(bp.double().sum())
    assert torch.equal(v_band, v_full)
    assert abs(float(sum(sums)) - float(tot.double().sum())) < 1e-6 * abs(float(tot.double().sum()))
    # -- whole steps
    def run(pipe):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        m = DepthGaussianSplattingModel(ModelConfig(sh_degree=3, sh_degree_interval=0, pipeline_ssim=pipe,
                                                    pipeline_ssim_min_tiles=0), params)
        for i in range(3):
            m.train_step(views[i % 2])
        torch.cuda.synchronize()
        return m
    a, b = run(True), run(False)
    assert a._side_stream is not None and getattr(b, "_side_stream", None) is None
    assert torch.equal(a.params.flat, b.params.flat) and torch.equal(a.optimizer.exp_avg_sq, b.optimizer.exp_avg_sq)
    assert torch.equal(a.last["tile_loss"], b.last["tile_loss"])
    la = a.loss_from(a.last["tile_loss"], a.last["ssim_sum"], views[0])
    lb = b.loss_from(b.last["tile_loss"], b.last["ssim_sum"], views[0])
    assert abs(float(la["main_loss"]) - float(lb["main_loss"])) < 1e-6 * abs(float(lb["main_loss"]))


def _nerfstudio_stub():
    """A stand-in for the parts of the nerfstudio API the plugin shell touches (shape per SURVEY App. A /
    reference legacy/config_tactile.py:23-56, legacy/model_tactile.py:38-56): config containers that keep
    their keyword arguments, `Model` = nn.Module that stores config / kwargs and calls populate_modules(),
    `Cameras` with batched intrinsics and camera_to_worlds.  NOT nerfstudio -- it only lets the shell's own
    code execute."""
    import dataclasses, sys, types

    class Cfg:
        def __init__(self, *a, **kw):
            self.__dict__.update(kw)

    @dataclasses.dataclass
    class NSModelConfig:
        _target: type = None

        def setup(self, **kw):
            return self._target(self, **kw)

    class Model(torch.nn.Module):
        def __init__(self, config, scene_box=None, num_train_data=0, **kwargs):
            super().__init__()
            self.config, self.scene_box, self.num_train_data, self.kwargs = config, scene_box, num_train_data, kwargs
            self.populate_modules()

    class Cameras:
        def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height):
            self.camera_to_worlds = camera_to_worlds
            t = lambda v: torch.tensor([[v]])
            self.fx, self.fy, self.cx, self.cy, self.width, self.height = t(fx), t(fy), t(cx), t(cy), t(width), t(height)

    # -- data side: a stand-in for nerfstudio's own nerfstudio-data parser (what the plugin's parser extends):
    #    frames sorted by file_path, train/eval split by fraction, poses centred and scaled to +-1, the depth
    #    file list + unit factor in metadata (the behaviour of reference legacy/dataparser_tactile.py:150-312)
    @dataclasses.dataclass
    class NerfstudioDataParserConfig:
        _target: type = None
        data: str = None
        depth_unit_scale_factor: float = 1e-3
        train_split_fraction: float = 0.9
        load_3D_points: bool = False

        def setup(self):
            return self._target(self)

    class DataparserOutputs(Cfg):
        pass

    class BatchCameras:
        def __init__(self, c2w, fx, fy, cx, cy, W, H):
            n = c2w.shape[0]
            self.camera_to_worlds = c2w
            f = lambda v: torch.full((n, 1), float(v))
            self.fx, self.fy, self.cx, self.cy = f(fx), f(fy), f(cx), f(cy)
            self.width, self.height = torch.full((n, 1), W), torch.full((n, 1), H)

        def __getitem__(self, i):
            return Cameras(self.camera_to_worlds[i:i + 1], float(self.fx[i]), float(self.fy[i]), float(self.cx[i]),
                           float(self.cy[i]), int(self.width[i]), int(self.height[i]))

    class Nerfstudio:
        def __init__(self, config):
            self.config = config

        def get_dataparser_outputs(self, split="train"):
            return self._generate_dataparser_outputs(split)

        def _generate_dataparser_outputs(self, split="train"):
            import json, math, os
            root = str(self.config.data)
            meta = json.load(open(os.path.join(root, "transforms.json")))
            frames = sorted(meta["frames"], key=lambda fr: fr["file_path"])
            n = len(frames)
            n_train = math.ceil(n * self.config.train_split_fraction)
            i_train = np.linspace(0, n - 1, n_train, dtype=int)
            idx = i_train if split == "train" else np.setdiff1d(np.arange(n), i_train)
            c2w = torch.tensor(np.stack([np.asarray(fr["transform_matrix"], np.float64) for fr in frames]))
            centre = c2w[:, :3, 3].mean(0)
            c2w[:, :3, 3] -= centre
            scale = 1.0 / float(c2w[:, :3, 3].abs().max())
            c2w[:, :3, 3] *= scale
            sel = [frames[i] for i in idx]
            return DataparserOutputs(
                dataparser_transform=torch.cat([torch.eye(3, dtype=torch.float64), -centre[:, None]], 1).float(),
                image_filenames=[os.path.join(root, fr["file_path"]) for fr in sel],
                cameras=BatchCameras(c2w[idx][:, :3, :].float(), meta["fl_x"], meta["fl_y"], meta["cx"], meta["cy"], meta["w"], meta["h"]),
                dataparser_scale=scale,
                metadata={"depth_filenames": [os.path.join(root, fr["depth_file_path"]) for fr in sel] if "depth_file_path" in sel[0] else None,
                          "depth_unit_scale_factor": self.config.depth_unit_scale_factor})

    class InputDataset(torch.utils.data.Dataset):
        exclude_batch_keys_from_device = ["image", "mask"]

        def __init__(self, dataparser_outputs, scale_factor=1.0):
            self._dataparser_outputs = dataparser_outputs
            self.metadata = dataparser_outputs.metadata
            self.cameras = dataparser_outputs.cameras

        def __len__(self):
            return len(self._dataparser_outputs.image_filenames)

        def get_metadata(self, data):
            return {}

        def __getitem__(self, i):
            from PIL import Image
            img = torch.from_numpy(np.asarray(Image.open(self._dataparser_outputs.image_filenames[i]).convert("RGB"),
                                              dtype=np.float32) / 255.0)
            data = {"image_idx": i, "image": img}
            data.update(self.get_metadata(data))
            return data

    class FullImageDatamanager:
        dataset_type = InputDataset

        def __class_getitem__(cls, item):
            return type(f"FullImageDatamanager[{item.__name__}]", (cls,), {"dataset_type": item})

        def __init__(self, config):
            self.config = config
            self.train_dataparser_outputs = config.dataparser.setup().get_dataparser_outputs("train")
            self.train_dataset = self.dataset_type(self.train_dataparser_outputs)

        def next_train(self, step):
            i = step % len(self.train_dataset)
            return self.train_dataset.cameras[i], self.train_dataset[i]

    class FullImageDatamanagerConfig(Cfg):
        def setup(self):
            return self._target(self)

    import enum

    class TrainingCallbackLocation(enum.Enum):
        BEFORE_TRAIN_ITERATION = 1
        AFTER_TRAIN_ITERATION = 2

    class TrainingCallback:
        def __init__(self, where_to_run, func, update_every_num_iters=None, iters=None, args=None, kwargs=None):
            self.where_to_run, self.func, self.args, self.kwargs = where_to_run, func, args or [], kwargs or {}

        def run_callback_at_location(self, step, location):
            if location in self.where_to_run:
                self.func(*self.args, **self.kwargs, step=step)

    layout = {"nerfstudio.configs.base_config": dict(ViewerConfig=Cfg),
              "nerfstudio.data.datasets.base_dataset": dict(InputDataset=InputDataset),
              "nerfstudio.engine.callbacks": dict(TrainingCallback=TrainingCallback, TrainingCallbackLocation=TrainingCallbackLocation),
              "nerfstudio.engine.trainer": dict(TrainerConfig=Cfg),
              "nerfstudio.plugins.types": dict(MethodSpecification=Cfg),
              "nerfstudio.cameras.cameras": dict(Cameras=Cameras),
              "nerfstudio.data.datamanagers.full_images_datamanager": dict(FullImageDatamanagerConfig=FullImageDatamanagerConfig,
                                                                           FullImageDatamanager=FullImageDatamanager),
              "nerfstudio.data.dataparsers.nerfstudio_dataparser": dict(NerfstudioDataParserConfig=NerfstudioDataParserConfig,
                                                                        Nerfstudio=Nerfstudio),
              "nerfstudio.engine.optimizers": dict(AdamOptimizerConfig=Cfg),
              "nerfstudio.engine.schedulers": dict(ExponentialDecaySchedulerConfig=Cfg),
              "nerfstudio.models.base_model": dict(Model=Model, ModelConfig=NSModelConfig),
              "nerfstudio.pipelines.base_pipeline": dict(VanillaPipelineConfig=Cfg)}
    mods = {}
    for name, attrs in layout.items():
        parts = name.split(".")
        for i in range(1, len(parts) + 1):
            mods.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
        mods[name].__dict__.update(attrs)
    return mods, Cameras


def test_nerfstudio_plugin_shell_executes_against_a_stub(dev):
    """nerfstudio cannot be installed here, so the MethodSpecification / Model shell of
    touch_gs_amd/nerfstudio_plugin.py never ran.  This test runs it against a minimal stand-in for the
    nerfstudio API (see _nerfstudio_stub): the module imports, the method spec carries the reference's
    method name and the three flags, the Model builds its parameter groups from seed points, and a
    nerfstudio-style iteration (get_outputs -> get_loss_dict -> backward -> one Adam per group with the
    spec's learning rates) reduces the loss."""
    import importlib, sys
    import touch_gs_amd.nerfstudio_plugin as plug
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    mods, Cameras = _nerfstudio_stub()
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        plug = importlib.reload(plug)
        assert plug.available
        spec = plug.depth_gaussian_splatting
        assert spec.config.method_name == "depth-gaussian-splatting"
        mc = spec.config.pipeline.model
        assert (mc.depth_loss_mult, mc.depth_loss_type, mc.uncertainty_weight) == (0.2, "DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 1.0)
        assert set(spec.config.optimizers) == set(plug.PARAM_GROUP_LRS)
        assert spec.config.optimizers["xyz"]["scheduler"].lr_final == plug.XYZ_LR_FINAL
        N, W, H = 3000, 128, 80
        view = make_view(N, W, H, 3, 5, dev)
        P, _ = synthetic_gaussians(N, W, H, 3, 99)
        mc.depth_loss_mult, mc.uncertainty_weight = 0.005, 0.01          # scripts/train_bunny_real.sh:52
        assert (mc.num_downscales, mc.resolution_schedule) == (2, 250)   # Splatfacto's defaults
        mc.num_downscales = 0                                            # (the schedule is exercised by the on-disk test below)
        mc.random_fill = 0                                               # this test starts from exactly its seed points
        model = mc.setup(scene_box=None, num_train_data=1, seed_points=(P["means"], torch.rand(N, 3) * 255))
        groups = model.get_param_groups()
        assert set(groups) == set(plug.PARAM_GROUP_LRS)
        assert {k for k, _ in model.named_parameters()} == {f"gauss_params.{k}" for k in groups}
        # the stub camera is what nerfstudio hands over: OpenGL camera-to-world [1,3,4] + batched intrinsics
        c2w = torch.linalg.inv(torch.tensor(view.cam.viewmat, dtype=torch.float64).reshape(4, 4))
        c2w[:3, 1:3] *= -1                                              # OpenCV -> OpenGL axes
        cams = Cameras(c2w[None, :3, :].float(), view.cam.fx, view.cam.fy, view.cam.cx, view.cam.cy, W, H)
        opts = {k: torch.optim.Adam(v, lr=spec.config.optimizers[k]["optimizer"].lr * 20, eps=1e-15) for k, v in groups.items()}
        batch = {"image": view.rgb, "depth_image": view.depth[..., None], "uncertainty": view.uncertainty[..., None]}
        losses = []
        for it in range(10):
            out = model.get_outputs(cams)
            assert out["rgb"].shape == (H, W, 3) and out["depth"].shape == (H, W, 1) and out["accumulation"].shape == (H, W, 1)
            loss = sum(model.get_loss_dict(out, batch).values())
            for o in opts.values():
                o.zero_grad()
            loss.backward()
            for o in opts.values():
                o.step()
            losses.append(float(loss))
        assert losses[-1] < 0.95 * losses[0], losses
        assert "psnr" in model.get_metrics_dict(model.get_outputs(cams), batch)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        importlib.reload(plug)


def test_nerfstudio_plugin_trains_with_touch_supervision_from_disk(dev, tmp_path):
    """The plugin's data path and refinement callbacks, executed against the stand-in API on a scene
    directory in the reference's format: the method spec's datamanager (FullImageDatamanager[TactileDepthDataset]
    over TactileDataParser) delivers batch["depth_image"] and batch["uncertainty"] in scene units (uint16 mm x
    1e-3 x dataparser scale, legacy/dataparser_tactile.py:65-66,229-235,301-312), so the tactile loss is
    applied (depth_loss > 0); a nerfstudio-style loop that runs the model's training callbacks around
    every iteration densifies the Gaussians (N changes, the Adam states follow)."""
    import importlib, sys
    import touch_gs_amd.nerfstudio_plugin as plug
    from touch_gs_amd.scene import make_view
    N, W, H, deg = 3000, 96, 64, 1
    root = tmp_path / "scene"
    _write_reference_scene(root, dev, N, W, H, deg)
    mods, Cameras = _nerfstudio_stub()
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        plug = importlib.reload(plug)
        spec = plug.depth_gaussian_splatting
        dmc = spec.config.pipeline.datamanager
        dmc.dataparser.data = str(root)
        dmc.dataparser.train_split_fraction = 0.8
        assert dmc.dataparser.uncertainty_floor == 0.0     # reference parity by default; the few-view preset's value:
        dmc.dataparser.uncertainty_floor = 0.05
        dm = dmc.setup()
        assert type(dm.train_dataset).__name__ == "TactileDepthDataset" and len(dm.train_dataset) == 5
        outs = dm.train_dataparser_outputs
        assert outs.metadata["uncertainty_filenames"][1].endswith("fused_output_dir_uncertainty/1.png")
        cam0, batch0 = dm.next_train(0)
        assert batch0["depth_image"].shape == (H, W, 1) and batch0["uncertainty"].shape == (H, W, 1)
        v0 = make_view(N, W, H, deg, 11, dev, view=0, n_views=6)
        sc = outs.dataparser_scale
        assert torch.allclose(batch0["depth_image"][..., 0], v0.depth.cpu() * sc, atol=1e-3 * sc + 1e-6)   # mm quantisation
        # (the dataparser's uncertainty floor -- 0.05 in the map's own units under the few-view preset, DESIGN 10 item 6 -- bounds the touch : vision
        # weight ratio of the depth loss; the map the model sees is the floored one)
        floor = dmc.dataparser.uncertainty_floor
        assert floor == 0.05 and (v0.uncertainty < floor).any()
        assert torch.allclose(batch0["uncertainty"][..., 0], v0.uncertainty.cpu().clamp(min=floor) * sc, atol=1e-3 * sc + 1e-6)
        assert (batch0["depth_image"] == 0).float().mean() > 0.2          # unsupervised pixels stay 0

        mc = spec.config.pipeline.model
        mc.sh_degree, mc.sh_degree_interval = deg, 4
        mc.num_downscales, mc.resolution_schedule = 1, 4                 # Splatfacto's schedule: half resolution for 4 steps
        mc.depth_loss_mult, mc.uncertainty_weight = 0.005, 0.01          # scripts/train_bunny_real.sh:52
        mc.warmup_length, mc.refine_every, mc.densify_grad_thresh = 4, 4, 1e-7   # refine early, on any gradient
        # the dataparser delivers the scene's touch point cloud (points_touch.npy / points_colors.npy, reference
        # utils/create_point_cloud_from_touches.py:243-244) in ITS frame, and the pipeline hands it to the model as
        # seed_points -- exactly what nerfstudio's VanillaPipeline does with metadata["points3D_xyz" / "points3D_rgb"]
        md = outs.metadata
        seeds = torch.from_numpy(np.load(root / "points_touch.npy")).float()
        centre = torch.tensor(np.stack([np.asarray(f["transform_matrix"])[:3, 3] for f in
                                        __import__("json").load(open(root / "transforms.json"))["frames"]]).mean(0)).float()
        assert md["points3D_xyz"].shape == (200, 3) and md["points3D_rgb"].dtype == torch.uint8
        assert torch.allclose(md["points3D_xyz"], (seeds - centre) * sc, atol=1e-5)
        assert torch.equal(md["points3D_rgb"], torch.from_numpy(np.load(root / "points_colors.npy")).float().clamp(0, 255).to(torch.uint8))
        assert (mc.random_fill, mc.max_seed_points) == (50000, 50000)   # defaults: touch cloud (subsampled) + random fill of the cube
        filled = mc.setup(scene_box=None, num_train_data=5, seed_points=(md["points3D_xyz"], md["points3D_rgb"]))
        assert filled.gaussians.num_points == 200 + 50000
        assert torch.allclose(filled.gaussians.params["xyz"].detach().cpu()[:200], md["points3D_xyz"], atol=1e-6)
        del filled
        mc.random_fill = 0
        model = mc.setup(scene_box=None, num_train_data=5, seed_points=(md["points3D_xyz"], md["points3D_rgb"]))
        assert model.gaussians.num_points == 200          # random_fill = 0: the model starts from exactly the touch points
        assert torch.allclose(model.gaussians.params["xyz"].detach().cpu(), md["points3D_xyz"], atol=1e-6)
        dc = model.gaussians.params["features_dc"].detach().cpu()[:, 0] * 0.28209479177387814 + 0.5
        assert torch.allclose(dc, md["points3D_rgb"].float() / 255.0, atol=1e-5)
        groups = model.get_param_groups()
        opts = {k: torch.optim.Adam(v, lr=spec.config.optimizers[k]["optimizer"].lr * 10, eps=1e-15) for k, v in groups.items()}

        class Attrs:   # TrainingCallbackAttributes: the callbacks reach the optimizers through .optimizers.optimizers
            class optimizers:
                pass
        Attrs.optimizers.optimizers = opts
        callbacks = model.get_training_callbacks(Attrs)
        Loc = sys.modules["nerfstudio.engine.callbacks"].TrainingCallbackLocation
        n0, depth_losses, seen_n, degs = model.gaussians.num_points, [], set(), []
        # 14 iterations: Splatfacto pauses splitting and culling for num_train_data + refine_every = 9 steps after a reset
        # boundary (step 0 is one: densify.py `refinement_after`), so the first refinement that changes N is the one at step 12
        for step in range(14):
            for cb in callbacks:
                cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
            cam, batch = dm.next_train(step)
            batch = {k: (v.to(dev) if hasattr(v, "to") else v) for k, v in batch.items()}
            out = model.get_outputs(cam)
            d = 2 if step < 4 else 1                                    # resolution schedule (supervision follows the render)
            assert out["rgb"].shape == (H // d, W // d, 3) and out["depth"].shape == (H // d, W // d, 1), (step, out["rgb"].shape)
            ld = model.get_loss_dict(out, batch)
            depth_losses.append(float(ld["depth_loss"]))
            degs.append(model.gaussians.active_sh_degree())
            for o in opts.values():
                o.zero_grad()
            sum(ld.values()).backward()
            for o in opts.values():
                o.step()
            for cb in callbacks:
                cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
            seen_n.add(model.gaussians.num_points)
        assert all(d > 0 for d in depth_losses), depth_losses          # the touch supervision reaches the loss
        assert degs[0] == 0 and degs[-1] == 1                           # SH ramp driven by the step callback
        info = model.refiner.last_info
        assert info is not None and info["after"] != info["before"] and len(seen_n) > 1, (info, seen_n)
        n1 = model.gaussians.num_points
        assert n1 != n0
        # the registered parameters, the optimizer's parameter list and its Adam state all follow the refinement
        for k, o in opts.items():
            p = model.gaussians.params[k]
            assert p.shape[0] == n1 and o.param_groups[0]["params"][0] is p and model.gauss_params[k] is p
            st = o.state[p]
            assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape
        assert len({id(q) for o in opts.values() for q in o.state}) == 6     # no stale state entries
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        importlib.reload(plug)


@pytest.mark.parametrize("deg,interval", [(3, 0), (1, 0), (3, 3)])
def test_color_prefetch_is_bit_identical(dev, deg, interval):
    """Colour prefetch (tgs_project_bwd_adam_next -> tgs_project_bin_sort_colors): the optimizer
    kernel of step t evaluates the colours the UPDATED Gaussians show to the camera of step t+1 and
    that step's K1 takes them instead of reading the SH rows.  Same parameters, Adam moments and
    images, bit for bit, as the plain step sequence -- also when the announced next view is not the
    one that comes (host-side mismatch), when the active SH degree changes in between (interval 3:
    degrees 0,0,0,1,1,1,...), and when the optimizer kernel was skipped on an overflowed frame (the
    tag word keeps its old value and K1 falls back to the SH rows on the device)."""
    from touch_gs_amd import ops
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H = 4100, 160, 96
    views = [make_view(N, W, H, 3, 7, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, 3, 99)

    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=interval), params)

    order = [0, 1, 2, 3, 1, 1, 0, 2]
    plain, pre = fresh(), fresh()
    used = 0
    for i, v in enumerate(order):
        plain.train_step(views[v])
        # announce the true next view, except at i == 3 where a wrong one is announced
        nxt = views[order[i + 1]] if i + 1 < len(order) else None
        if i == 3:
            nxt = views[3]
        ready = getattr(pre, "_prefetch_ready", None)
        used += int(ready is not None and ready.matches(views[v].cam, N, pre.active_sh_degree()))
        pre.train_step(views[v], next_view=nxt)
        assert torch.equal(pre.last["rgb"], plain.last["rgb"]), i
        assert torch.equal(pre.last["splats"], plain.last["splats"]), i
    assert used >= (4 if interval == 0 else 2), used      # the prefetched colours really were taken
    for x, y in ((pre.params.flat, plain.params.flat), (pre.optimizer.exp_avg, plain.optimizer.exp_avg),
                 (pre.optimizer.exp_avg_sq, plain.optimizer.exp_avg_sq)):
        assert torch.equal(x, y)

    # device-side fallback: the frame of the announcing step overflows (capacity 16 pairs), its guarded
    # optimizer kernel does nothing, so the announced colours were never written
    m = fresh()
    m.train_step(views[0], next_view=views[1])
    good = m.budget
    m.budget = ops.IntersectBudget(capacity=16, sync=False)
    flat = m.params.flat.clone()
    m.train_step(views[1], next_view=views[2])           # overflows: no update, no prefetch
    assert int(m.last["status"][1]) == 1 and torch.equal(m.params.flat, flat)
    m.optimizer.t -= 1; m.step -= 1                      # the step did not happen
    m.budget = good
    ready = m._prefetch_ready
    assert ready is not None and ready.matches(views[2].cam, N, m.active_sh_degree())   # the host would use it
    assert int(ready.tag_word) != ready.tag                                              # the device says no
    ref = fresh()
    ref.train_step(views[0]); ref.train_step(views[2])
    m.train_step(views[2])
    assert torch.equal(m.last["rgb"], ref.last["rgb"]) and torch.equal(m.params.flat, ref.params.flat)


@pytest.mark.parametrize("densify,speculative", [(False, False), (True, False), (False, True)])
def test_front_prefetch_is_bit_identical(dev, densify, speculative):
    """Front prefetch (tgs_project_bwd_adam_next_front -> tgs_project_bin_sort_front): the fused optimizer kernel of
    step t also runs K1 of step t+1's view -- records, radii, pair ranges, tile counts -- on the parameters it has
    just updated; step t+1 only scans, fills and sorts.  Records, images, parameters and moments bit for bit equal to
    the same sequence with the prefetch switched off, with the synchronous and the sync-free budget, with radii
    requested (densification statistics), when the announced view is not the one that comes, and across an
    overflowing frame (the voided optimizer kernel leaves the tag word alone: the next frame is voided by its scan launch)."""
    from touch_gs_amd import ops
    from touch_gs_amd.densify import DensifyConfig
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 4100, 160, 96, 3
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)

    def fresh(front):
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
        m.front_prefetch = front
        if densify:
            m.enable_densification(DensifyConfig(warmup_length=1000, refine_every=1000))   # statistics only
        if speculative:
            m.enable_speculative_budget(capacity=200_000)
        return m

    order = [0, 1, 2, 3, 1, 1, 0, 2]
    on, off = fresh(True), fresh(False)
    taken = 0
    for i, v in enumerate(order):
        nxt = views[order[i + 1]] if i + 1 < len(order) else None
        if i == 3:
            nxt = views[3]                                   # wrong announcement: the front buffers are dropped
        ready = getattr(on, "_prefetch_ready", None)
        use = ready is not None and ready.front_issued and ready.matches(views[v].cam, N, deg)
        taken += int(use)
        on.train_step(views[v], next_view=nxt)
        off.train_step(views[v], next_view=nxt)
        assert not use or ready.front is None                # consumed by this step's front half
        assert torch.equal(on.last["splats"], off.last["splats"]), i
        assert torch.equal(on.last["rgb"], off.last["rgb"]), i
        assert torch.equal(on.last["status"], off.last["status"]), i
        if densify:
            assert torch.equal(on.last["radii"], off.last["radii"]), i
    assert taken >= 5, taken
    assert getattr(off, "_prefetch_ready", None) is None or off._prefetch_ready.front is None
    if speculative:
        on.flush(); off.flush()
    for x, y in ((on.params.flat, off.params.flat), (on.optimizer.exp_avg, off.optimizer.exp_avg),
                 (on.optimizer.exp_avg_sq, off.optimizer.exp_avg_sq)):
        assert torch.equal(x, y)
    if densify:
        assert torch.equal(on.density.grad_norm_sum, off.density.grad_norm_sum)
        assert torch.equal(on.density.max_radius, off.density.max_radius)

    if not densify and not speculative:
        # an overflowing frame between two announced steps, SAME budget object (sync-free, fixed capacity): the fused
        # kernel of the overflowing step is voided, the sticky word voids the next frame too -- the scan launch of its
        # front finish sees the tag mismatch and empties the frame (no K1 is re-run for it); nothing is updated
        m = fresh(True)
        m.budget = ops.IntersectBudget(capacity=16, sync=False)
        flat = m.params.flat.clone()
        m.train_step(views[0], next_view=views[1])           # overflows: no update, front of view 1 announced
        ready = m._prefetch_ready
        assert ready is not None and ready.front_issued and int(ready.tag_word) != ready.tag
        m.train_step(views[1], next_view=views[2])           # front path taken on the host, voided on the device
        assert ready.front is None and int(m.last["status"][1]) == 1
        assert torch.equal(m.params.flat, flat)
        with pytest.raises(RuntimeError):
            m.budget.check()


def test_spatial_sort_is_a_pure_relayout(dev):
    """model.spatial_sort(): Morton order of parameters + Adam moments.  The render is unchanged and a
    few train steps give the same model up to the permutation (only the order of exactly equal depths
    inside a tile list can differ: ties break by Gaussian id)."""
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians
    N, W, H, deg = 20000, 320, 200, 3
    views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(2)]
    P, _ = synthetic_gaussians(N, W, H, deg, 99)
    def fresh():
        params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
        return DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    a, b = fresh(), fresh()
    for step in range(2):            # some Adam state before the sort
        a.train_step(views[step % 2]); b.train_step(views[step % 2])
    perm = b.spatial_sort()
    assert torch.equal(b.params.means, a.params.means[perm]) and torch.equal(b.optimizer.exp_avg_sq[:3 * N].view(N, 3),
                                                                             a.optimizer.exp_avg_sq[:3 * N].view(N, 3)[perm])
    # neighbours in memory are neighbours in space
    d_sorted = (b.params.means[1:] - b.params.means[:-1]).norm(dim=1).median()
    d_before = (a.params.means[1:] - a.params.means[:-1]).norm(dim=1).median()
    assert d_sorted < 0.2 * d_before
    ra, rb = a.get_outputs(views[0].cam), b.get_outputs(views[0].cam)
    assert torch.allclose(ra["rgb"], rb["rgb"], atol=2e-6) and torch.allclose(ra["depth_acc"], rb["depth_acc"], atol=2e-5)
    for step in range(2, 5):
        a.train_step(views[step % 2]); b.train_step(views[step % 2])
    d = (b.params.flat[:3 * N].view(N, 3) - a.params.flat[:3 * N].view(N, 3)[perm]).abs()
    assert float(torch.quantile(d.flatten().float(), 0.999)) < 1e-5


_DP4_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
dp = parallel.init_from_env(backend="gloo")           # 4 ranks share the single GPU
assert dp.world == 4
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
N, W, H, deg = 3000, 128, 80, 3
views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=4) for v in range(4)]     # configs[3]: a batch of 4 views
P, _ = synthetic_gaussians(N, W, H, deg, 99)
def fresh():
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    m.spatial_sort()
    return m
m = fresh()
for step in range(2):
    m.train_step(views[dp.views_for_step(step, 4)], dp)
torch.cuda.synchronize()
dp.assert_replicas_identical(m.params.flat)
ref = fresh()
for step in range(2):
    ref.optimizer.lrs["means"] = ref.config.lr_means_at(step)
    g = torch.zeros_like(ref.params.grad)
    for r in range(4):
        ref.forward_backward(views[r])
        g += ref.params.grad
    ref.params.grad.copy_(g * 0.25)
    ref.optimizer.step()
torch.cuda.synchronize()
# a sum over 4 ranks is not associative-order identical to the sequential reference: a 1-ulp difference
# in a near-zero gradient can flip an Adam step (eps = 1e-15), so the comparison is statistical
dabs = (ref.params.flat - m.params.flat).abs()
d = torch.quantile(dabs[::3].float(), 0.999).item()
assert d < 2e-6 and dabs.max().item() < 1e-3, (d, dabs.max().item())
dp.barrier()
if dp.rank == 0: print("DP4_OK", d)
'''


_DP_TRANSPORT_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, steps = (int(v) for v in sys.argv[3:7])
deg = 3
# the ranks share the single GPU of the test box over gloo; on a box with a GPU per rank they take one each over RCCL
# (and the peer transport's stores cross xGMI for real)
multi = torch.cuda.device_count() >= int(os.environ["WORLD_SIZE"])
dp = parallel.init_from_env(backend="nccl" if multi else "gloo")
dev = torch.device("cuda", dp.local_rank if multi else 0); torch.cuda.set_device(dev)
views = [make_view(N, W, H, deg, 7, dev, view=v, n_views=8) for v in range(8)]
P, _ = synthetic_gaussians(N, W, H, deg, 99)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
m.spatial_sort()
m.enable_speculative_budget()
dp.color_chunks = int(os.environ.get("CHUNKS", "2"))
first = None
for step in range(steps):
    m.train_step(views[dp.views_for_step(step, 8)], dp, next_view=views[dp.views_for_step(step + 1, 8)])
    if step == 0:
        m.flush()
        first = m.params.flat.cpu()
m.flush()
dp.check_transport()
torch.cuda.synchronize()
dp.assert_replicas_identical(m.params.flat)
if dp.transport == "ipc":
    assert dp.peer is not None and dp.peer.seq == steps and dp.peer.bytes_pushed > 0
if dp.rank == 0:
    torch.save(dict(flat=m.params.flat.cpu(), first=first, geom_end=m.optimizer.geom_end()), sys.argv[2])
dp.barrier()
if dp.peer is not None: dp.peer.close()
if dp.rank == 0: print("DPT_OK", dp.transport, flush=True)
'''


@pytest.mark.parametrize("world,N,W,H,safe", [(2, 20000, 320, 208, "0"), (4, 20000, 320, 208, "0"), (3, 4001, 160, 96, "0"),
                                               (2, 20000, 320, 208, "1")])
def test_peer_transport_equals_collective_transport(dev, tmp_path, world, N, W, H, safe):
    """TGS_DP_TRANSPORT=ipc (csrc/peer.hip: the factored exchange as direct stores into IPC-mapped peer buffers --
    colour blocks pushed to every rank's gather slots, geometry gradients as a direct reduce-scatter + rank-order sum
    + all-gather, flag words with system-scope release / acquire) against the collective form, with 2 / 4 / 3
    processes sharing the one GPU of the test box (IPC handles work across processes on one device; on a node
    the same stores cross xGMI).  Three steps of the trainer's data-parallel form (2 row chunks, sync-free budget,
    DP front prefetch).  The SH update of a step only depends on the gathered colour blocks (a copy): after the first
    step the SH segment is bit-identical for any number of ranks.  The geometry segment sums `world` gradients: a + b
    is order-free, so 2 ranks stay bit-identical throughout; with more ranks the rank-order sum and gloo's tree differ
    by rounding (statistical agreement from then on, as between the collective form and the hand-averaged single
    process).  Replicas are bit-identical in every case.  N = 4001: block sizes that are not multiples of 16 bytes
    (scalar path of the push kernel).  safe = "1": TGS_PEER_SAFE_FLAGS, the flags raised by a launch of their own."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_transport_worker.py"
    script.write_text(_DP_TRANSPORT_WORKER)
    out = {}
    for i, transport in enumerate(("rccl", "ipc")):
        port = free_port()
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   TGS_DP_TRANSPORT=transport, CHUNKS="2", TGS_PEER_SAFE_FLAGS=safe)
        f = tmp_path / f"{transport}.pt"
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                            "--master-addr", "127.0.0.1", "--master-port", port, str(script), root, str(f),
                            str(N), str(W), str(H), "3"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and f"DPT_OK {transport}" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
        out[transport] = torch.load(f)
    a, b, ge = out["rccl"]["flat"], out["ipc"]["flat"], out["rccl"]["geom_end"]
    assert torch.equal(out["rccl"]["first"][ge:], out["ipc"]["first"][ge:])     # SH rows after step 1: any number of ranks
    if world == 2:
        assert torch.equal(a, b) and torch.equal(out["rccl"]["first"], out["ipc"]["first"])
    else:
        d = (a - b).abs()
        assert float(torch.quantile(d[::5].float(), 0.999)) < 2e-6 and float(d.max()) < 0.11, (float(d.max()),)


_DP_FULL_WORKER = r'''
import json, os, sys, time, torch
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import parallel
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, seed, steps = (int(v) for v in sys.argv[2:7])
deg = 3
# all ranks share the single GPU of the test box over gloo; with a GPU per rank they take one each over RCCL
multi = torch.cuda.device_count() >= int(os.environ["WORLD_SIZE"])
dp = parallel.init_from_env(backend="nccl" if multi else "gloo")
world = dp.world
dev = torch.device("cuda", dp.local_rank if multi else 0); torch.cuda.set_device(dev)
views = [make_view(N, W, H, deg, seed, dev, view=v, n_views=8) for v in range(8)]
P, _ = synthetic_gaussians(N, W, H, deg, seed)
def fresh():
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
    m.spatial_sort()
    return m
m = fresh()
m.enable_speculative_budget()                         # the trainer's default: no read-back of the intersection count
assert dp.color_chunks == (4 if world >= 4 else 1)    # pipelined exchange: the trainer's default from 4 ranks on
t0 = time.time()
for step in range(steps):                             # the rank's next view is announced: front prefetch of the DP step
    m.train_step(views[dp.views_for_step(step, 8)], dp, next_view=views[dp.views_for_step(step + 1, 8)])
m.flush()
dp.check_transport()                                  # (peer transport: no wait timed out)
torch.cuda.synchronize()
t_dp = time.time() - t0
dp.assert_replicas_identical(m.params.flat)           # bit for bit on every rank
assert (dp.peer is not None) == (os.environ.get("TGS_DP_TRANSPORT") == "ipc")
stats = None
if dp.rank == 0:
    ref = fresh()                                     # a single process that averages the ranks' views by hand
    for step in range(steps):
        ref.optimizer.lrs["means"] = ref.config.lr_means_at(step)
        g = torch.zeros_like(ref.params.grad)
        for r in range(world):
            ref.forward_backward(views[(step * world + r) % 8])
            g += ref.params.grad
        ref.params.grad.copy_(g / world)
        ref.optimizer.step()
    torch.cuda.synchronize()
    d = (ref.params.flat - m.params.flat).abs()
    moved = (ref.params.flat - fresh().params.flat).abs() > 0
    stats = dict(world=world, N=N, replays=getattr(m, "speculative_replays", 0), seconds=round(t_dp, 1),
                 moved=float(moved.float().mean()), q999=float(torch.quantile(d[::max(7, d.numel() // 8_000_000)].float(), 0.999)),
                 frac_gt_1e5=float((d > 1e-5).float().mean()), max=float(d.max()),
                 front_prefetch=bool(m._prefetch_ready is not None and m._prefetch_ready.front_issued))
dp.barrier()
if stats is not None: print("DPFULL " + json.dumps(stats), flush=True)
'''


# (the 8 x 5 M instance runs for the peer transport only: the collective form is covered at 4 x 1 M, and the pair
# costs a minute of the driver's GPU-test budget -- VERDICT r4 item 8)
@pytest.mark.parametrize("world,N,W,H,seed,transport", [(4, 1_000_000, 1920, 1080, 1236, "rccl"),
                                                        (4, 1_000_000, 1920, 1080, 1236, "ipc"),
                                                        (8, 5_000_000, 3840, 2160, 1238, "ipc")])
def test_data_parallel_fullsize(dev, tmp_path, world, N, W, H, seed, transport):
    """BASELINE configs[3] (a batch of 4 views of the 1 M / 1080p scene on 4 ranks) and the data-parallel part of
    configs[4] (5 M Gaussians, SH 3, 4K, 8 ranks) at FULL size: the ranks share the one GPU of the test box over gloo
    (the exchange logic, chunking and kernels are the ones that run over RCCL; only the transport differs).  Two
    steps of the trainer's default data-parallel form -- pipelined factored exchange in 4 row chunks, sync-free
    intersection budget, the rank's next view announced (front prefetch inside the geometry Adam) -- leave every
    replica bit-identical and agree with a single process that averages the ranks' gradients by hand -- with the
    collective transport ("rccl": torch.distributed, here gloo) and with the peer transport ("ipc": direct stores into
    IPC-mapped buffers, csrc/peer.hip).  The
    agreement is statistical, as in the miniature: a sum over ranks is not order-identical to the sequential sum and
    Adam's first steps are lr * sign(g), so a 1-ulp difference in a near-zero gradient flips a whole step."""
    import json, os, subprocess, sys
    if torch.cuda.device_count() < world and torch.cuda.mem_get_info()[1] < 150e9:
        pytest.skip("needs a 288 GB MI355X: 8 replicas of the 5 M scene share the device")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp_full_worker.py"
    script.write_text(_DP_FULL_WORKER)
    # the ranks share this box's ONE GPU with the pytest process: hand back what its allocator has cached from the tests
    # before (8 x 5 M replicas need ~160 GB of the 288)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    print("pytest process before the ranks start: reserved %.1f GB, free on the device %.1f GB" %
          (torch.cuda.memory_reserved() / 1e9, torch.cuda.mem_get_info()[0] / 1e9))
    torch.cuda.empty_cache()
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
               TGS_DP_TRANSPORT=transport)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), root,
                        str(N), str(W), str(H), str(seed), "2"],
                       env=env, capture_output=True, text=True, timeout=2400)
    line = [l for l in r.stdout.splitlines() if l.startswith("DPFULL ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-4000:]
    st = json.loads(line[-1][7:])
    print(st)
    assert st["world"] == world and st["front_prefetch"] and st["moved"] > 0.5
    # measured (r4): 4 x 1 M: q99.9 of |difference| 3.7e-9, 1.0e-7 of the entries differ by more than 1e-5, max 8.6e-5;
    # 8 x 5 M: 3.0e-8, 2.3e-7, 2.8e-4.  (A flipped first Adam step would be 2 lr <= 0.1, opacity logit.)
    assert st["q999"] < 1e-6 and st["frac_gt_1e5"] < 1e-5 and st["max"] < 0.11, st


def test_data_parallel_four_ranks_batch_of_four(dev, tmp_path):
    """BASELINE configs[3] in miniature: a batch of 4 views, one per rank, 4 ranks (sharing the one GPU
    of the test box, gloo): the factored exchange equals a single process that averages the 4 views'
    gradients, replicas stay identical (Morton-ordered rows on every rank)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "dp4_worker.py"
    script.write_text(_DP4_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), root],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP4_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
