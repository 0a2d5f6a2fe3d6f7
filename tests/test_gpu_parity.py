"""GPU parity tests: every HIP kernel against the CPU oracle on identical seeded inputs.

The oracle is the build's own restatement of the published algorithm (parity UNPINNED against the
reference rasterizer, whose source is absent -- see oracle/__init__.py).  Tolerances: 1e-4
relative on RGB/depth (BASELINE.json north_star) for every pixel whose threshold tests are not
decision-ambiguous under fp32 rounding; integer outputs exact.
"""
import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests.util import amd_cam, clamp_scene, count_clamped, rect_from, relerr, scene, splat_fields, to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _project(dev, P, cam, deg):
    from touch_gs_amd import ops
    D = to_dev(P, dev)
    sp, radii = ops.project_fwd(amd_cam(cam), D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"],
                                deg, want_radii=True)
    return D, sp, radii


@pytest.mark.parametrize("N,W,H,deg,seed", [(2000, 160, 96, 3, 1), (500, 80, 48, 0, 2), (3000, 200, 120, 2, 3),
                                            (1000, 100, 70, 1, 4)])
def test_project_fwd(dev, N, W, H, deg, seed):
    P, cam = scene(N, W, H, deg, seed)
    _, sp, radii = _project(dev, P, cam, deg)
    f = splat_fields(sp, radii)
    pr = O.project(P["means"], P["log_scales"], P["quats"], P["opac_logit"], P["sh"], cam, deg)
    v = pr["valid"]
    # the packed rect is a subset of the normative B.4 rect, and visible <=> radius > 0
    assert torch.equal(f["visible"], f["radius"] > 0)
    rr, orr = f["rect"], pr["rect"]
    sub = (f["hits"] == 0) | ((rr[:, 0] >= orr[:, 0]) & (rr[:, 1] >= orr[:, 1]) & (rr[:, 2] <= orr[:, 2]) & (rr[:, 3] <= orr[:, 3]))
    assert sub[f["radius"] == pr["radius"]].all()
    # integer decisions: radius may differ only where 3*sqrt(lam1) sits within fp32 noise of an integer
    same = (f["radius"] == pr["radius"])
    assert same.float().mean() > 0.995, f"radius mismatch fraction {1 - same.float().mean():.4f}"
    both = v & same & (f["radius"] > 0)
    assert both.sum() > 0.5 * v.sum()
    for k in ("conic", "rgb"):
        e = relerr(f[k][both], pr[k][both], floor=1e-3)
        assert e.max() < TOL, (k, e.max())
    # pixel coordinates: 1e-4 relative to the image scale (fp32 ulp at 2000 px is already 1.2e-4 px)
    assert relerr(f["xy"][both], pr["xy"][both], floor=float(max(W, H))).max() < 1e-5
    assert relerr(f["depth"][both], pr["depth"][both]).max() < 1e-5
    assert relerr(f["opac"], pr["opac"]).max() < 1e-5
    assert relerr(f["rgb"], pr["rgb"], floor=1e-3).max() < TOL  # colour defined for every Gaussian


@pytest.mark.parametrize("N,W,H,seed,morton", [(3000, 160, 96, 5, False), (800, 50, 35, 6, False),
                                                 (20000, 320, 200, 7, False), (20000, 320, 200, 8, True),
                                                 (100000, 800, 800, 9, True), (6000, 320, 200, 10, "big")])
def test_bin_sort_exact(dev, N, W, H, seed, morton):
    """morton=True: Gaussians in 3-D Morton order, so that a binning group's tile bounding box is
    small and K3a takes the LDS-aggregated counting path (one global atomic per (group, tile))."""
    from touch_gs_amd import ops
    from touch_gs_amd.optim import morton_order
    P, cam = scene(N, W, H, 0, seed)
    if morton == "big":   # > 1024 pairs per group inside a small bounding box: aggregated + direct counting mixed
        P["log_scales"] += 1.2
    if morton:
        perm = morton_order(P["means"])
        P = {k: v[perm].contiguous() for k, v in P.items()}
    _, sp, _ = _project(dev, P, cam, 0)
    gb, ts, sg, st = ops.bin_sort(amd_cam(cam), sp)
    f = splat_fields(sp)
    gid, tstart = O.bin_and_sort(f["rect"], f["hits"] > 0, f["depth"], cam)
    n, ovf = st.tolist()
    assert ovf == 0 and n == len(gid)
    assert np.array_equal(ts.cpu().numpy().astype(np.int64), tstart)
    assert np.array_equal(sg[:n].cpu().numpy().astype(np.int64), gid)


@pytest.mark.parametrize("N,W,H,seed,scale_up", [(100000, 800, 800, 9, 0.0), (30000, 1280, 720, 21, 1.5)])
def test_pair_allocator_when_xcd_regions_fill_up(dev, N, W, H, seed, scale_up):
    """The pair index space is cut into one region per XCD; a group whose own region is full takes its range from
    another one.  With the capacity anywhere between the frame's pair count and the sufficient bound status[2], every
    frame must EITHER report the overflow OR produce exactly the lists of a generously sized run -- on buffers that
    come out of the allocator full of garbage.  (Rounds 3-4 gave a failed attempt's pairs back with an atomicSub; a
    third group's range could then overlap a second one's, the lost pairs left unwritten list entries -- garbage
    Gaussian ids -- and K6 faulted on a held-out view of an 8-view model at 89 % of the capacity.)"""
    from touch_gs_amd import ops
    from touch_gs_amd.optim import morton_order
    P, cam = scene(N, W, H, 0, seed)
    P["log_scales"] += scale_up
    perm = morton_order(P["means"])
    P = {k: v[perm].contiguous() for k, v in P.items()}
    _, sp, _ = _project(dev, P, cam, 0)
    acam = amd_cam(cam)
    big = ops.IntersectBudget()
    gb0, ts0, sg0, st0 = ops.bin_sort(acam, sp, big)
    n, need = big.last_n, big.last_need
    ref_ts, ref_sg = ts0.clone(), sg0[:n].clone()
    ran_clean = 0
    for frac in (0.0, 0.02, 0.05, 0.1, 0.2, 0.35, 0.5, 0.75, 1.0):
        cap = int(n + frac * (need - n)) + (1 if frac == 0.0 else 0)
        for rep in range(3):
            junk = [torch.full((k,), 0x7fffffff, dtype=torch.int32, device=dev) for k in (cap, 2 * cap, 4 * cap + 64, 1 << 16)]
            del junk
            b = ops.IntersectBudget(capacity=cap, sync=False)
            gb, ts, sg, st = ops.bin_sort(acam, sp, b)
            cnt, ovf = st.tolist()
            if ovf:
                assert int(ts[-1]) == 0            # an overflowed frame has empty lists
                continue
            ran_clean += 1
            assert cnt == n
            assert torch.equal(ts, ref_ts)
            assert torch.equal(sg[:n], ref_sg), f"cap {cap} ({frac} of the way from n={n} to need={need})"
    assert ran_clean >= 3     # cap = need always fits


def _blend_inputs(dev, N, W, H, deg, seed, clamp=False, **kw):
    from touch_gs_amd import ops
    P, cam = (clamp_scene if clamp else scene)(N, W, H, deg, seed, **kw)
    D, sp, _ = _project(dev, P, cam, deg)
    acam = amd_cam(cam)
    gb, ts, sg, st = ops.bin_sort(acam, sp)
    n = int(ts[-1])      # total length of the tile lists
    return P, cam, acam, D, sp, gb, ts, sg, n


@pytest.mark.parametrize("N,W,H,deg,seed,clamp", [(3000, 160, 96, 3, 11, False), (600, 50, 35, 1, 12, False),
                                                   (8000, 256, 144, 0, 13, False), (2000, 128, 80, 2, 14, True)])
def test_rasterize_fwd(dev, N, W, H, deg, seed, clamp):
    from touch_gs_amd import ops
    P, cam, acam, D, sp, gb, ts, sg, n = _blend_inputs(dev, N, W, H, deg, seed, clamp=clamp)
    rgb, depth, fT, fidx = ops.rasterize_fwd(acam, sp, sg, ts, want_idx=True)
    rgb2, depth2, fT2, none_idx = ops.rasterize_fwd(acam, sp, sg, ts)
    assert none_idx is None and torch.equal(rgb, rgb2) and torch.equal(depth, depth2) and torch.equal(fT, fT2)
    f = splat_fields(sp)
    out = O.blend(f["xy"], f["conic"], f["opac"], f["rgb"], f["depth"],
                  sg[:n].cpu().numpy().astype(np.int64), ts.cpu().numpy().astype(np.int64), cam, want_margin=True)
    clear = (out["margin"] > 1e-3).numpy()
    assert clear.mean() > 0.9
    er = relerr(rgb.cpu().numpy(), out["rgb"].numpy(), floor=1e-2)
    ed = relerr(depth.cpu().numpy(), out["depth_acc"].numpy(), floor=1e-2)
    eT = np.abs(fT.cpu().numpy() - out["final_T"].numpy())
    assert er[clear].max() < TOL and ed[clear].max() < TOL and eT[clear].max() < TOL
    assert np.array_equal(fidx.cpu().numpy()[clear], out["final_idx"].numpy()[clear])
    # ambiguous pixels may flip one threshold decision: bounded by one alpha_min-level contribution
    assert np.abs(rgb.cpu().numpy() - out["rgb"].numpy()).max() < 0.02


@pytest.fixture
def k7_quad_everywhere():
    """K7's four-waves-per-tile form (k_raster_bwd_quad) for every tile that walks more than 8 entries of (almost) any
    frame, instead of only in chain-bound frames (the rule of tgs_set_k7_quad)."""
    from touch_gs_amd import ops
    before = ops.set_k7_quad()
    ops.set_k7_quad(1, 8)
    yield
    ops.set_k7_quad(*before)


@pytest.mark.parametrize("quad", [False, True])
@pytest.mark.parametrize("N,W,H,deg,seed,clamp", [(2000, 128, 80, 3, 21, False), (500, 50, 35, 0, 22, False),
                                                   (2000, 128, 80, 3, 23, True), (2500, 64, 64, 1, 24, True)])
def test_rasterize_bwd(dev, N, W, H, deg, seed, clamp, quad, request):
    """clamp=True: a third of the Gaussians have opacity > 0.999 and e^2 larger axes, so that
    thousands of (pixel, Gaussian) pairs sit ON the alpha = 0.999 clamp, whose gradient App. B.7
    passes through (K7: raster.hip `q *= max(exp2(-s)/0.999, 1)`).  quad: the same through K7's four-wave form."""
    from touch_gs_amd import ops
    if quad:
        request.getfixturevalue("k7_quad_everywhere")
    P, cam, acam, D, sp, gb, ts, sg, n = _blend_inputs(dev, N, W, H, deg, seed, clamp=clamp)
    rgb, depth, fT, fidx = ops.rasterize_fwd(acam, sp, sg, ts)
    g = torch.Generator().manual_seed(seed)
    v_rgb = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    v_d = torch.randn(H, W, generator=g, dtype=torch.float64)
    v_a = torch.randn(H, W, generator=g, dtype=torch.float64)
    partials, _ = ops.rasterize_bwd(acam, sp, gb, sg, ts, rgb, depth, fT,
                                    v_rgb.float().to(dev), v_d.float().to(dev), v_a.float().to(dev))
    v = ops.reduce_partials(acam, sp, gb, partials).cpu().double()
    f = splat_fields(sp)
    if clamp:
        n_cl = count_clamped(f, sg[:n].cpu().numpy(), ts.cpu().numpy(), cam)
        assert n_cl >= 1000, n_cl
    leaves = {k: f[k].clone().requires_grad_(True) for k in ("xy", "conic", "opac", "rgb", "depth")}
    out = O.blend(leaves["xy"], leaves["conic"], leaves["opac"], leaves["rgb"], leaves["depth"],
                  sg[:n].cpu().numpy().astype(np.int64), ts.cpu().numpy().astype(np.int64), cam, want_margin=True)
    L = (out["rgb"] * v_rgb).sum() + (out["depth_acc"] * v_d).sum() + (out["alpha"] * v_a).sum()
    L.backward()
    frac_clear = (out["margin"] > 1e-3).double().mean().item()
    ref = dict(xy=leaves["xy"].grad, depth=leaves["depth"].grad, opac=leaves["opac"].grad,
               conic=leaves["conic"].grad, rgb=leaves["rgb"].grad)
    got = dict(xy=v[:, 0:2], depth=v[:, 2], opac=v[:, 3], conic=v[:, 4:7], rgb=v[:, 7:10])
    for k in ref:
        scale = ref[k].abs().max().item()
        err = (got[k] - ref[k]).abs()
        # a Gaussian's gradient sums many pixels: compare against the per-tensor scale; pixels with
        # ambiguous threshold decisions (1 - frac_clear of them) perturb a few entries
        bad = (err > 1e-3 * scale + 1e-4 * ref[k].abs()).double().mean().item()
        assert bad < 0.02 + 5 * (1 - frac_clear), (k, bad, frac_clear)
        assert np.median(relerr(got[k].numpy(), ref[k].numpy(), floor=1e-3 * scale)) < 3e-5, k


@pytest.mark.parametrize("rule", [(8, 16), (1, 8)])
def test_k7_quad_form_on_an_object_centric_scene(dev, rule):
    """An object-centric scene (80 % of the Gaussians in the central tenth of the image) is chain-bound: a few tiles walk
    hundreds of entries while the frame's balanced load is a few dozen per wave slot.  K7 then gives every tile that walks
    more than min_walk entries to k_raster_bwd_quad (default rule (8, 16); (1, 8): almost every tile).  Against one wave per tile: the same tile losses bit for
    bit (k_raster_bwd computes them for every tile), every (tile, Gaussian) partial record within rounding of the
    record's own magnitude -- the quadrant totals are added in a different order, nothing else differs -- with the fused
    loss of the train step (L1 + tactile depth / uncertainty) and with plain upstream gradients; poisoned partial
    buffers show that every record of every listed pair is written by exactly one of the two kernels."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg = 60_000, 640, 400, 3
    P, intr = synthetic_gaussians(N, W, H, deg, 5, clustered=True)
    D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
    cam = make_camera(intr, 1, 8, bg=(0.1, 0.2, 0.3))
    sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
    rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    n = (ts[1:cam.num_tiles + 1] - ts[:cam.num_tiles]).long()
    I = int(n.sum())
    TW, TH = (W + 15) // 16, (H + 15) // 16
    pad = torch.zeros(TH * 16, TW * 16, dtype=torch.int64, device=dev)
    pad[:H, :W] = fT.stop_pos.long().clamp(max=int(n.max()))
    tmax = torch.minimum(pad.view(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(TH * TW, 256).max(1).values, n)
    assert int(tmax.max()) * 8192 > int(tmax.sum()) * rule[0], (int(tmax.max()), int(tmax.sum()))   # chain-bound by tgs_set_k7_quad's rule
    n_long = int((tmax > rule[1]).sum())
    assert 0 < n_long < cam.num_tiles, (n_long, cam.num_tiles, I, int(tmax.max()))
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    gd = (2 + 4 * torch.rand(H, W, generator=g)).to(dev)
    gd[torch.rand(H, W, generator=g).to(dev) < 0.3] = 0.0
    unc = (0.001 + 5 * torch.rand(H, W, generator=g)).to(dev)
    loss = dict(gt_rgb=gt, l1_weight=0.8 / (3 * H * W), gt_depth=gd, depth_weight=0.2 / (H * W), uncertainty=unc,
                uncertainty_weight=1.0, eps=1e-6)
    v_rgb = torch.randn(H, W, 3, generator=g).to(dev)
    v_d, v_a = torch.randn(H, W, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    modes = {"loss": dict(v_rgb=v_rgb * 1e-6, loss=loss, want_tile_loss=True),          # the fused train step's form
             "plain": dict(v_rgb=v_rgb, v_depth=v_d, v_alpha=v_a)}
    before = ops.set_k7_quad()
    try:
        out = {}
        for name, (f, mw) in (("one", (0, 192)), ("quad", rule)):
            ops.set_k7_quad(f, mw)
            for mode, kw in modes.items():
                buf = torch.full((sg.shape[0], 12), float("nan"), device=dev)
                partials, tl = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, partials=buf, **kw)
                out[name, mode] = (partials[:, :10].clone(), None if tl is None else tl.clone())
    finally:
        ops.set_k7_quad(*before)
    for mode in modes:
        a, ta = out["one", mode]
        b, tb = out["quad", mode]
        wa, wb = ~torch.isnan(a).any(dim=1), ~torch.isnan(b).any(dim=1)
        assert torch.equal(wa, wb) and int(wa.sum()) == I      # every listed pair's record written, nothing else
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        if ta is not None:
            assert torch.equal(ta, tb)
        a, b = a[wa], b[wa]
        assert not torch.equal(a, b)                                    # the long tiles did take the other kernel
        mag = a.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)       # a record's own magnitude
        rel = ((a - b).abs() / mag).amax(dim=1)
        # (the moment terms of a record are differences of sums 10 - 100x their size: tile-centred second moments)
        q99 = float(torch.quantile(rel[::max(1, rel.numel() // 4_000_000)], 0.99))
        print(f"k7 quad vs one wave, {mode}: {n_long} long tiles of {cam.num_tiles}, max {float(rel.max()):.2e}, q99 {q99:.2e}")
        assert float(rel.max()) < 3e-4 and q99 < 1e-5, (float(rel.max()), q99)


@pytest.mark.parametrize("N,W,H,deg,seed,clamp", [(20000, 320, 208, 1, 51, False), (3000, 100, 70, 0, 52, True), (30000, 256, 256, 3, 53, False)])
def test_k7_block_form_matches_the_one_wave_form_and_the_oracle(dev, N, W, H, deg, seed, clamp):
    """k_raster_bwd_blocks (TgsRasterOpts.k7_blocks = 1; round 6, measured and NOT the default -- DESIGN 5.1e): the
    backward with a different Gaussian in every 16-lane DPP row.  Same decisions as the forward, sums formed in another
    order: every (tile, Gaussian) record within rounding of the one-wave form's record (its own magnitude as the scale),
    NaN-poisoned buffers show that exactly the listed pairs are written, tile losses equal up to rounding, and the ten
    screen-space gradients hold the statistical bar of test_rasterize_bwd against the fp64 oracle."""
    from touch_gs_amd import ops
    P, cam, acam, D, sp, gb, ts, sg, n = _blend_inputs(dev, N, W, H, deg, seed, clamp=clamp)
    rgb, depth, fT, _ = ops.rasterize_fwd(acam, sp, sg, ts)
    g = torch.Generator().manual_seed(seed)
    v_rgb = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    v_d = torch.randn(H, W, generator=g, dtype=torch.float64)
    v_a = torch.randn(H, W, generator=g, dtype=torch.float64)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    loss = dict(gt_rgb=gt, l1_weight=0.8 / (3 * H * W))
    out = {}
    for name, opts in (("one", ops.raster_opts(k7_quad=0, k7_blocks=0)), ("blocks", ops.raster_opts(k7_blocks=1))):
        buf = torch.full((sg.shape[0], 12), float("nan"), device=dev)
        partials, tl = ops.rasterize_bwd(acam, sp, gb, sg, ts, rgb, depth, fT, v_rgb.float().to(dev), v_d.float().to(dev),
                                         v_a.float().to(dev), loss=loss, want_tile_loss=True, partials=buf, opts=opts)
        out[name] = (partials[:, :10].clone(), tl.clone(),
                     ops.reduce_partials(acam, sp, gb, torch.nan_to_num(partials)).cpu().double())
    (a, ta, _), (b, tb, v) = out["one"], out["blocks"]
    wa, wb = ~torch.isnan(a).any(dim=1), ~torch.isnan(b).any(dim=1)
    # (tile losses: the same 256 pixel terms per tile, summed over another lane -> pixel mapping: equal up to rounding)
    assert torch.equal(wa, wb) and int(wa.sum()) == n and torch.allclose(ta, tb, rtol=1e-5, atol=1e-9)
    a, b = a[wa], b[wa]
    mag = a.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
    rel = ((a - b).abs() / mag).amax(dim=1)
    print(f"k7 block form vs one wave: max {float(rel.max()):.2e}, q99 {float(torch.quantile(rel[:: max(1, rel.numel() // 1_000_000)], 0.99)):.2e}")
    assert float(rel.max()) < 1e-3 and float(torch.quantile(rel[:: max(1, rel.numel() // 1_000_000)], 0.99)) < 2e-5, float(rel.max())
    f = splat_fields(sp)
    leaves = {k: f[k].clone().requires_grad_(True) for k in ("xy", "conic", "opac", "rgb", "depth")}
    o = O.blend(leaves["xy"], leaves["conic"], leaves["opac"], leaves["rgb"], leaves["depth"],
                sg[:n].cpu().numpy().astype(np.int64), ts.cpu().numpy().astype(np.int64), cam, want_margin=True)
    l1 = (0.8 / (3 * H * W)) * (o["rgb"] - gt.cpu().double()).abs().sum()
    ((o["rgb"] * v_rgb).sum() + (o["depth_acc"] * v_d).sum() + (o["alpha"] * v_a).sum() + l1).backward()
    frac_clear = (o["margin"] > 1e-3).double().mean().item()
    ref = dict(xy=leaves["xy"].grad, depth=leaves["depth"].grad, opac=leaves["opac"].grad, conic=leaves["conic"].grad, rgb=leaves["rgb"].grad)
    got = dict(xy=v[:, 0:2], depth=v[:, 2], opac=v[:, 3], conic=v[:, 4:7], rgb=v[:, 7:10])
    for k in ref:
        scale = ref[k].abs().max().item()
        bad = ((got[k] - ref[k]).abs() > 1e-3 * scale + 1e-4 * ref[k].abs()).double().mean().item()
        assert bad < 0.02 + 5 * (1 - frac_clear), (k, bad, frac_clear)
        assert np.median(relerr(got[k].numpy(), ref[k].numpy(), floor=1e-3 * scale)) < 3e-5, k


def test_k7_quad_form_without_a_schedule_with_an_empty_first_tile_and_dirty_slot_counters(dev):
    """ADVICE r4 (medium) regression: the four-wave launch draws its tiles from 8 slot counters behind the tile starts.
    They are zeroed by block 0 of the one-wave launch that precedes it -- BEFORE that block's early returns: with no
    tile_order (spatial schedule) block 0 owns tile 0, which is EMPTY in most object-centric frames, and a return above
    the store left whatever the allocator had put there (the four-wave workgroups then drew slots beyond the schedule:
    no tile of theirs was ever composited).  Here: tile 0 is empty by construction, no tile_order, the eight counter
    words hold 0x7fffffff before every backward, every tile walking more than 8 entries is the four-wave launch's.
    Against the one-wave form: every listed pair's record written exactly once (NaN-poisoned buffers), equal up to the
    rounding of the four-term quadrant sum; and the ten screen-space gradients against the fp64 oracle."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg = 30_000, 480, 320, 1
    P, intr = synthetic_gaussians(N, W, H, deg, 9, clustered=True)
    D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
    cam = make_camera(intr, 1, 8, bg=(0.1, 0.2, 0.3))
    T = cam.num_tiles
    for _ in range(2):       # make tile 0 empty: whoever reaches it becomes invisible (opacity <= 1/255 emits no pairs)
        sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
        first = sg[int(ts[0]):int(ts[1])].long()
        D["opac_logit"][first] = -20.0
    assert int(ts[1]) - int(ts[0]) == 0 and int(ts[T]) > 50_000
    rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    # the same buffer (starts + the rasterizer's scratch words) WITHOUT the schedule that rides on the tensor object
    ts_plain = ts[:]
    assert getattr(ts_plain, "tile_order", None) is None and ops._tile_start_len(ts_plain) == ops._tile_start_len(ts)
    whole = torch.empty(0, dtype=torch.int32, device=dev).set_(ts.untyped_storage(), ts.storage_offset(), (T + 513,))
    slot_words = [T + 1 + 64 * x + 32 for x in range(8)]          # TGS_SLOTCTR_AT(T, x), tgs_common.h
    g = torch.Generator().manual_seed(4)
    v_rgb = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    v_d, v_a = torch.randn(H, W, generator=g, dtype=torch.float64), torch.randn(H, W, generator=g, dtype=torch.float64)
    out = {}
    for name, opts in (("one", ops.raster_opts(k7_quad=0)), ("quad", ops.raster_opts(k7_quad=1, k7_quad_min_walk=8))):
        whole[slot_words] = 0x7fffffff
        buf = torch.full((sg.shape[0], 12), float("nan"), device=dev)
        partials, _ = ops.rasterize_bwd(cam, sp, gb, sg, ts_plain, rgb, depth, fT, v_rgb.float().to(dev), v_d.float().to(dev),
                                        v_a.float().to(dev), partials=buf, opts=opts)
        out[name] = partials[:, :10].clone()
        out[name + "_v"] = ops.reduce_partials(cam, sp, gb, torch.nan_to_num(partials)).cpu().double()
    a, b = out["one"], out["quad"]
    wa, wb = ~torch.isnan(a).any(dim=1), ~torch.isnan(b).any(dim=1)
    assert torch.equal(wa, wb) and int(wa.sum()) == int(ts[T])            # every listed pair written, by either form
    assert not torch.equal(a[wa], b[wa])                                    # the long tiles did take the four-wave kernel
    mag = a[wa].abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
    rel = ((a[wa] - b[wa]).abs() / mag).amax(dim=1)
    assert float(rel.max()) < 3e-4 and float(torch.quantile(rel[:: max(1, rel.numel() // 1_000_000)], 0.99)) < 1e-5, float(rel.max())
    # ... and against the oracle (same statistical bar as test_rasterize_bwd)
    f = splat_fields(sp)
    n = int(ts[T])
    leaves = {k: f[k].clone().requires_grad_(True) for k in ("xy", "conic", "opac", "rgb", "depth")}
    ocam = O.Camera(viewmat=torch.from_numpy(np.asarray(cam.viewmat, np.float64)), fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy,
                    W=W, H=H, bg=cam.bg)
    o = O.blend(leaves["xy"], leaves["conic"], leaves["opac"], leaves["rgb"], leaves["depth"],
                sg[:n].cpu().numpy().astype(np.int64), ts.cpu().numpy().astype(np.int64), ocam, want_margin=True)
    ((o["rgb"] * v_rgb).sum() + (o["depth_acc"] * v_d).sum() + (o["alpha"] * v_a).sum()).backward()
    frac_clear = (o["margin"] > 1e-3).double().mean().item()
    v = out["quad_v"]
    ref = dict(xy=leaves["xy"].grad, depth=leaves["depth"].grad, opac=leaves["opac"].grad, conic=leaves["conic"].grad, rgb=leaves["rgb"].grad)
    got = dict(xy=v[:, 0:2], depth=v[:, 2], opac=v[:, 3], conic=v[:, 4:7], rgb=v[:, 7:10])
    for k in ref:
        scale = ref[k].abs().max().item()
        bad = ((got[k] - ref[k]).abs() > 1e-3 * scale + 1e-4 * ref[k].abs()).double().mean().item()
        assert bad < 0.02 + 5 * (1 - frac_clear), (k, bad, frac_clear)


def _records_in_list_order(cam, sp, gb, ts, sg, partials):
    """K7's partial records gathered in (tile, list position) order.  A record's address is its pair index, and the pair
    ranges of the binning groups are handed out in whatever order the workgroups arrive: two runs of the front half give
    the same LISTS but different addresses."""
    T = cam.num_tiles
    n_t = (ts[1:T + 1] - ts[:T]).long()
    tile = torch.repeat_interleave(torch.arange(T, device=sp.device), n_t)
    spv = sp.view(-1, 12)
    gid = sg[: int(ts[T])].long()
    rect = spv[gid, 10].contiguous().view(torch.int32)
    x0, y0, w_ = rect & 255, (rect >> 8) & 255, (rect >> 16) & 255
    TW = cam.tiles[0]
    P = gb.long()[gid // 256] + spv[gid, 11].contiguous().view(torch.int32).long() + (tile // TW - y0) * w_ + (tile % TW - x0)
    assert P.unique().numel() == P.numel()
    written = torch.zeros(partials.shape[0], dtype=torch.bool, device=sp.device)
    written[P] = True
    assert torch.equal(written, ~torch.isnan(partials[:, :10]).any(dim=1))      # exactly the listed pairs were written
    return partials[P, :10]


@pytest.mark.parametrize("scan_min", [64, 16])
def test_k7_scan_form_on_the_longest_tiles(dev, scan_min):
    """k_raster_bwd_scan (round 6): the longest tiles of a chain-bound frame with the batch's 64 ENTRIES in the lanes of a
    wave -- transmittance and the sum behind as DPP prefix scans, each lane accumulating its own entry's sums, 16 waves per
    tile -- beside the four-wave form for the rest.  Against one wave per tile on an object-centric scene, fused train-step
    loss and plain upstream gradients: exactly the listed pairs' records written (NaN-poisoned buffers), equal up to
    rounding of the record's own magnitude (records compared in list order: pair addresses differ between runs of the
    front half), tile losses bit for bit; and the scan form did take tiles (the result differs from the four-wave-only run)."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg = 60_000, 640, 400, 3
    P, intr = synthetic_gaussians(N, W, H, deg, 5, clustered=True)
    D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
    D["opac_logit"][::3] = 12.0                      # some opacities on the 0.999 clamp: both copies of the pixel loop run
    cam = make_camera(intr, 1, 8, bg=(0.1, 0.2, 0.3))
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    gd = (2 + 4 * torch.rand(H, W, generator=g)).to(dev)
    gd[torch.rand(H, W, generator=g).to(dev) < 0.3] = 0.0
    unc = (0.001 + 5 * torch.rand(H, W, generator=g)).to(dev)
    loss = dict(gt_rgb=gt, l1_weight=0.8 / (3 * H * W), gt_depth=gd, depth_weight=0.2 / (H * W), uncertainty=unc,
                uncertainty_weight=1.0, eps=1e-6)
    v_rgb = torch.randn(H, W, 3, generator=g).to(dev)
    v_d, v_a = torch.randn(H, W, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    modes = {"loss": dict(v_rgb=v_rgb * 1e-6, loss=loss, want_tile_loss=True), "plain": dict(v_rgb=v_rgb, v_depth=v_d, v_alpha=v_a)}
    before_q, before_s = ops.set_k7_quad(), ops.set_k7_scan()
    out = {}
    try:
        for name, (f, mw, sm) in (("one", (0, 192, 0)), ("quad", (8, 16, 0)), ("scan", (8, 16, scan_min))):
            ops.set_k7_quad(f, mw)
            ops.set_k7_scan(sm, 512)
            for mode, kw in modes.items():
                # fresh lists + ONE forward per backward: the walk statistics the chain-bound rule reads accumulate
                sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
                rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
                buf = torch.full((sg.shape[0], 12), float("nan"), device=dev)
                partials, tl = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, partials=buf, **kw)
                out[name, mode] = (_records_in_list_order(cam, sp, gb, ts, sg, partials), None if tl is None else tl.clone())
    finally:
        ops.set_k7_quad(*before_q)
        ops.set_k7_scan(*before_s)
    for mode in modes:
        a, ta = out["one", mode]
        b, tb = out["scan", mode]
        q, _ = out["quad", mode]
        assert a.shape == b.shape == q.shape and not torch.isnan(b).any()
        if ta is not None:
            assert torch.equal(ta, tb)
        assert not torch.equal(b, q)                                      # some tiles did take the scan form
        mag = a.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
        rel = ((a - b).abs() / mag).amax(dim=1)
        q99 = float(torch.quantile(rel[::max(1, rel.numel() // 4_000_000)], 0.99))
        print(f"k7 scan form vs one wave, {mode}: max {float(rel.max()):.2e}, q99 {q99:.2e}")
        assert float(rel.max()) < 1e-3 and q99 < 2e-5, (float(rel.max()), q99)


@pytest.mark.parametrize("N,W,H,deg,seed", [(1500, 128, 80, 3, 31), (400, 64, 48, 2, 32), (400, 64, 48, 0, 33)])
def test_project_bwd(dev, N, W, H, deg, seed):
    from touch_gs_amd import ops
    P, cam = scene(N, W, H, deg, seed)
    D, sp, radii = _project(dev, P, cam, deg)
    g = torch.Generator().manual_seed(seed)
    v_splats = torch.zeros(N, 12, dtype=torch.float64)
    v_splats[:, :10] = torch.randn(N, 10, generator=g, dtype=torch.float64)
    vm, vls, vq, vol, vsh, vxy = ops.project_bwd(amd_cam(cam), D["means"], D["log_scales"], D["quats"],
                                                 D["opac_logit"], D["sh"], deg, sp,
                                                 v_splats=v_splats.float().to(dev), want_v_xy=True)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    pr = O.project(Pg["means"], Pg["log_scales"], Pg["quats"], Pg["opac_logit"], Pg["sh"], cam, deg)
    f = splat_fields(sp, radii)
    vis = (f["radius"] > 0) & pr["valid"]
    vz = torch.where(vis[:, None], v_splats[:, :10], torch.zeros_like(v_splats[:, :10]))
    # geometry terms only flow for visible Gaussians; colour/opacity terms for all
    L = (pr["xy"] * vz[:, 0:2]).sum() + (pr["depth"] * vz[:, 2]).sum() + (pr["opac"] * v_splats[:, 3]).sum() \
        + (pr["conic"] * vz[:, 4:7]).sum() + (pr["rgb"] * v_splats[:, 7:10]).sum()
    L.backward()
    same = (f["radius"] > 0) == pr["valid"]
    pairs = [("means", vm), ("log_scales", vls), ("quats", vq), ("opac_logit", vol), ("sh", vsh)]
    for name, got in pairs:
        ref = Pg[name].grad[same]
        got = got.cpu().double()[same]
        scale = ref.abs().max().item()
        e = relerr(got.numpy(), ref.numpy(), floor=1e-4 * scale)
        assert np.quantile(e, 0.999) < 2e-3, (name, np.quantile(e, 0.999))
        assert np.median(e) < 1e-5, (name, np.median(e))
    assert torch.allclose(vxy.cpu().double(), v_splats[:, 0:2], atol=1e-6)


@pytest.mark.parametrize("clamp", [False, True])
def test_end_to_end_render_and_grads(dev, clamp):
    """Whole pipeline against the fp64 oracle; statistical because fp32 flips a few decisions.
    clamp=True puts 25 % of the Gaussians at opacity logit 12 (alpha clamp regime, B.7)."""
    from touch_gs_amd import ops
    N, W, H, deg = 4000, 192, 112, 3
    P, cam = (clamp_scene(N, W, H, deg, 42, frac=4, grow=2.0) if clamp else scene(N, W, H, deg, 41))
    D = to_dev(P, dev)
    for t in D.values():
        t.requires_grad_(True)
    rgb, depth, alpha, radii = ops.render(D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"],
                                          amd_cam(cam), deg)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out, pr, gid, ts = O.render(Pg["means"], Pg["log_scales"], Pg["quats"], Pg["opac_logit"], Pg["sh"], cam, deg,
                                want_margin=True)
    if clamp:
        assert count_clamped(pr, gid, ts, cam) >= 1000
    er = relerr(rgb.detach().cpu().numpy(), out["rgb"].detach().numpy(), floor=1e-2)
    ed = relerr(depth.detach().cpu().numpy(), out["depth_acc"].detach().numpy(), floor=1e-2)
    assert np.quantile(er, 0.99) < TOL and np.quantile(ed, 0.99) < TOL, (np.quantile(er, 0.99), np.quantile(ed, 0.99))
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(H, W, 3, generator=g, dtype=torch.float64)
    dgt = torch.rand(H, W, generator=g, dtype=torch.float64) * 5
    dgt[torch.rand(H, W, generator=g) < 0.3] = 0
    unc = torch.rand(H, W, generator=g, dtype=torch.float64) * 5 + 1e-3
    Lr = O.train_loss(out, gt, dgt, unc, ssim_lambda=0.0, depth_loss_mult=0.2, uncertainty_weight=1.0)
    Lr.backward()
    o = dict(rgb=rgb.double(), depth_acc=depth.double(), alpha=alpha.double())
    Lg = O.train_loss(o, gt.to(dev), dgt.to(dev), unc.to(dev), ssim_lambda=0.0, depth_loss_mult=0.2,
                      uncertainty_weight=1.0)
    Lg.backward()
    assert abs(Lg.item() - Lr.item()) < 1e-3 * abs(Lr.item())
    for k in D:
        ref, got = Pg[k].grad.numpy(), D[k].grad.cpu().double().numpy()
        scale = np.abs(ref).max()
        e = relerr(got, ref, floor=1e-3 * scale)
        assert np.quantile(e, 0.98) < 5e-3, (k, np.quantile(e, 0.98))
        cos = (ref * got).sum() / np.sqrt((ref * ref).sum() * (got * got).sum())
        assert cos > 0.9999, (k, cos)


def test_fused_loss_matches_autograd_path(dev):
    """K7's in-kernel L1 + depth/uncertainty loss == the same loss applied through autograd."""
    from touch_gs_amd import ops
    N, W, H, deg = 3000, 160, 96, 3
    P, cam = scene(N, W, H, deg, 51)
    acam = amd_cam(cam)
    D = to_dev(P, dev)
    g = torch.Generator().manual_seed(1)
    gt = torch.rand(H, W, 3, generator=g).to(dev)
    dgt = (torch.rand(H, W, generator=g) * 5)
    dgt[torch.rand(H, W, generator=g) < 0.3] = 0
    dgt = dgt.to(dev)
    unc = (torch.rand(H, W, generator=g) * 5 + 1e-3).to(dev)
    for loss_type, uw in (("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", 0.01), ("SIMPLE_LOSS", 1.0)):
        Dg = {k: v.clone().requires_grad_(True) for k, v in D.items()}
        rgb, depth, alpha, _ = ops.render(Dg["means"], Dg["log_scales"], Dg["quats"], Dg["opac_logit"], Dg["sh"], acam, deg)
        L = O.train_loss(dict(rgb=rgb, depth_acc=depth, alpha=alpha), gt, dgt, unc, ssim_lambda=0.0,
                         depth_loss_mult=0.3, depth_loss_type=loss_type, uncertainty_weight=uw)
        L.backward()
        sp = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
        gb, ts, sg, st = ops.bin_sort(acam, sp)
        r2, d2, fT, fidx = ops.rasterize_fwd(acam, sp, sg, ts)
        cnt = int((dgt > 0).sum())
        spec = dict(gt_rgb=gt, gt_depth=dgt, uncertainty=unc if loss_type != "SIMPLE_LOSS" else None,
                    l1_weight=1.0 / (3 * H * W), depth_weight=0.3 / cnt, uncertainty_weight=uw, eps=1e-6)
        partials, tl = ops.rasterize_bwd(acam, sp, gb, sg, ts, r2, d2, fT, loss=spec, want_tile_loss=True)
        vm, vls, vq, vol, vsh, _ = ops.project_bwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                   D["sh"], deg, sp, gb, partials)
        assert abs(tl.sum().item() - L.item()) < 1e-4 * abs(L.item())
        for name, got in (("means", vm), ("log_scales", vls), ("quats", vq), ("opac_logit", vol), ("sh", vsh)):
            ref = Dg[name].grad
            scale = ref.abs().max().item()
            assert (got - ref).abs().max().item() < 2e-4 * scale, (loss_type, name)


def test_slot_ok_bitmaps_do_not_change_the_backward(dev, monkeypatch):
    """TGS_SLOT_OK (off by default): K6 records which Gaussians changed the state of each 8x8 quadrant, K7
    skips the other (Gaussian, quadrant) evaluations -- the partial gradients are bit-identical."""
    from touch_gs_amd import ops
    P, cam = scene(20000, 320, 208, 3, 77)
    acam = amd_cam(cam)
    D = to_dev(P, dev)
    g = torch.Generator().manual_seed(3)
    v = [torch.randn(208, 320, 3, generator=g).to(dev), torch.randn(208, 320, generator=g).to(dev),
         torch.randn(208, 320, generator=g).to(dev)]
    res = []
    was = ops.set_raster_variant()
    ops.set_raster_variant(k7_front_to_back=True)     # the bitmaps belong to the front-to-back form of K7
    try:
        for on in (False, True):
            monkeypatch.setattr(ops, "SLOT_OK", on)
            sp, _, gb, ts, sg, st = ops.project_bin_sort(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 3)
            rgb, depth, fT, _ = ops.rasterize_fwd(acam, sp, sg, ts)
            assert (ts.slot_ok is not None) == on
            partials, _ = ops.rasterize_bwd(acam, sp, gb, sg, ts, rgb, depth, fT, *v)
            grads = ops.project_bwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 3, sp, gb, partials)
            res.append([rgb, depth, fT] + list(grads[:5]))
    finally:
        ops.set_raster_variant(k7_front_to_back=bool(was & 2))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    if res:   # some quadrant evaluations were really skipped
        words = ts.slot_ok.view(-1)
        assert int((words != 0).sum()) > 0


@pytest.mark.parametrize("N,W,H,seed,clamp", [(20000, 320, 208, 78, False), (3000, 100, 70, 79, True), (60000, 256, 256, 80, False)])
def test_k6_block_form_is_bit_identical(dev, monkeypatch, N, W, H, seed, clamp):
    """The 4x4-block form of K6 (every DPP row of the wave walks its own block's list, TGS_K6_BLOCKS=1) against
    the quadrant form: same eval / blend on the same pixels in the same list order -> images, final T and the
    last-contributor index equal bit for bit (a block wrongly culled would show up as a missing contribution)."""
    from touch_gs_amd import ops
    P, cam = (clamp_scene if clamp else scene)(N, W, H, 3, seed)
    acam = amd_cam(cam)
    D = to_dev(P, dev)
    sp, _, gb, ts, sg, st = ops.project_bin_sort(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 3)
    res = []
    was = ops.set_raster_variant()
    try:
        for on in (False, True):
            assert ops.set_raster_variant(k6_blocks=on) & 1 == int(on)
            out = ops.rasterize_fwd(acam, sp, sg, ts, want_idx=True)
            res.append(list(out) + [out[2].stop_pos])      # + the per-pixel stop positions the backward starts from
    finally:
        ops.set_raster_variant(k6_blocks=bool(was & 1))
    for a, b in zip(*res):
        assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()
    # stop position and last contributor are consistent: a pixel that stopped did so behind its last contributor
    fidx, stop = res[0][3], res[0][4]
    stopped = stop != 0x7fffffff
    assert bool((stop[stopped] > fidx[stopped]).all()) and bool(stopped.any())


@pytest.mark.parametrize("want_idx", [False, True])
def test_k6_quadrant_split_of_long_tiles_is_bit_identical(dev, want_idx):
    """The forward composites a tile with a long list with FOUR blocks of the same launch, one 8x8 quadrant each
    (raster_fwd_quadrant; tgs_set_k6_split): images, depth, final T, stop positions (and the last-contributor index)
    equal bit for bit to the unsplit launch, on an object-centric scene where the default rule splits the dense centre,
    with the rule forced onto every tile of the schedule's head (factor 1 at a tiny scene: lists > 256), and on a
    ragged image (width and height not multiples of 16).  The walk words K7's rule reads stay consistent: same maximum,
    a sum within the quarter-rounding of the split tiles."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    before = ops.set_k6_split()
    try:
        for N, W, H, factor in ((60_000, 640, 400, 4), (30_000, 250, 170, 1)):
            P, intr = synthetic_gaussians(N, W, H, 3, 5, clustered=True)
            D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
            cam = make_camera(intr, 1, 8, bg=(0.1, 0.2, 0.3))
            sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 3)
            T = cam.num_tiles
            n = (ts[1:T + 1] - ts[:T]).long()
            I = int(n.sum())
            n_split = int((n > max(256, (I * factor) >> 12)).sum())
            assert 0 < n_split < T, (n_split, T, I, int(n.max()))
            words = lambda: torch.as_strided(ts, (8, 2), (64, 1), T + 1).clone()     # (max, sum) per XCD behind the tile starts
            res, ww = [], []
            for f in (0, factor):
                ops.set_k6_split(f)
                torch.as_strided(ts, (8, 2), (64, 1), T + 1).zero_()                  # (the scan zeroes them once per frame)
                out = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=want_idx)
                res.append([o for o in out if o is not None] + [out[2].stop_pos])
                ww.append(words())
            for a, b in zip(*res):
                assert torch.equal(a, b), (a.float() - b.float()).abs().max().item()
            assert int(ww[0][:, 0].max()) == int(ww[1][:, 0].max()) > 0
            s0, s1 = int(ww[0][:, 1].sum()), int(ww[1][:, 1].sum())
            assert s0 > 0 and abs(s1 - s0) <= 0.5 * s0, (s0, s1)
    finally:
        ops.set_k6_split(before)


def test_deterministic_bitwise(dev):
    from touch_gs_amd import ops
    P, cam = scene(5000, 200, 120, 3, 61)
    acam = amd_cam(cam)
    D = to_dev(P, dev)
    res = []
    for _ in range(2):
        Dg = {k: v.clone().requires_grad_(True) for k, v in D.items()}
        rgb, depth, alpha, _ = ops.render(Dg["means"], Dg["log_scales"], Dg["quats"], Dg["opac_logit"], Dg["sh"], acam, 3)
        (rgb.sum() + depth.sum() + 0.5 * alpha.sum()).backward()
        res.append([rgb.detach().clone(), depth.detach().clone()] + [Dg[k].grad.clone() for k in Dg])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_edge_cases(dev):
    from touch_gs_amd import ops
    # (a) everything behind the camera -> background only, zero grads
    P, cam = scene(300, 70, 50, 1, 71, bg=(0.2, 0.4, 0.6))
    P["means"][:, 2] = -P["means"][:, 2] - 10
    D = {k: v.requires_grad_(True) for k, v in to_dev(P, dev).items()}
    rgb, depth, alpha, radii = ops.render(D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], amd_cam(cam), 1)
    assert int((radii > 0).sum()) == 0
    assert torch.allclose(rgb, torch.tensor([0.2, 0.4, 0.6], device=dev).expand(50, 70, 3))
    assert float(alpha.abs().max()) == 0 and float(depth.abs().max()) == 0
    (rgb.sum() + depth.sum()).backward()
    assert float(D["means"].grad.abs().max()) == 0
    # (b) one huge Gaussian covering every tile + many small ones in one tile (long list -> 2nd sort class)
    N = 6000
    P, cam = scene(N, 64, 48, 0, 72)
    P["means"][:] = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64) + 0.02 * torch.randn(N, 3, dtype=torch.float64)
    P["log_scales"][:] = -5.0
    P["log_scales"][0] = 1.0
    D = to_dev(P, dev)
    acam = amd_cam(cam)
    sp = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 0)
    gb, ts, sg, st = ops.bin_sort(acam, sp)
    f = splat_fields(sp)
    gid, tstart = O.bin_and_sort(f["rect"], f["hits"] > 0, f["depth"], cam)
    n = st.tolist()[0]
    assert (np.diff(tstart).max() > 2048), "test must exercise the large-list sort class"
    assert n == len(gid) and np.array_equal(sg[:n].cpu().numpy().astype(np.int64), gid)
    rgb, depth, fT, fidx = ops.rasterize_fwd(acam, sp, sg, ts)
    out = O.blend(f["xy"], f["conic"], f["opac"], f["rgb"], f["depth"], gid, tstart, cam, want_margin=True)
    clear = (out["margin"] > 1e-3).numpy()
    assert relerr(rgb.cpu().numpy(), out["rgb"].numpy(), floor=1e-2)[clear].max() < TOL


@pytest.mark.parametrize("N,lo,hi", [(40, 2, 64), (100, 64, 128), (200, 128, 256), (300, 256, 512), (700, 512, 1024),
                                       (1500, 1024, 2048), (3000, 2048, 4096), (5200, 4096, 16384)])
def test_sort_classes(dev, N, lo, hi):
    """Every list-length class of the per-tile sort (register-resident wave sort with 8 / 16 / 32 / 64
    keys per lane, then the LDS workgroup sort): order bit-exact vs the oracle, incl. depth ties."""
    from touch_gs_amd import ops
    P, cam = scene(N, 32, 32, 0, 90 + N)
    P["means"][:] = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64) + 0.01 * torch.randn(N, 3, dtype=torch.float64)
    P["means"][::7, 2] = 4.0          # exact depth ties: order must fall back to the Gaussian id
    P["log_scales"][:] = -6.0
    D = to_dev(P, dev)
    acam = amd_cam(cam)
    sp = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 0)
    gb, ts, sg, st = ops.bin_sort(acam, sp)
    f = splat_fields(sp)
    gid, tstart = O.bin_and_sort(f["rect"], f["hits"] > 0, f["depth"], cam)
    n = st.tolist()[0]
    assert lo < np.diff(tstart).max() <= hi, np.diff(tstart).max()
    assert n == len(gid) and np.array_equal(sg[:n].cpu().numpy().astype(np.int64), gid)


@pytest.mark.parametrize("N,hint,ok", [(700, 1024, True), (700, 0, True), (1500, 1024, False), (1500, 2000, True),
                                        (1500, 4096, True), (5200, 4096, False), (5200, 6000, True), (5200, -1, True)])
def test_max_list_hint_skips_sort_classes_or_voids_the_frame(dev, N, hint, ok):
    """tgs_bin_sort's max_list_hint (TGS_VERSION 300): the launches for list classes beyond the caller's bound are not
    issued.  A frame that keeps the promise is sorted bit for bit as without a hint; one that breaks it is VOID like an
    overflow -- status[1] = 1, sticky word raised, status[3] = the longest list -- and its long lists hold the tile's own
    ids in arrival order (valid indices: K6 / K7 behind it must not fault), never stale memory."""
    from touch_gs_amd import ops
    P, cam = scene(N, 32, 32, 0, 90 + N)
    P["means"][:] = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64) + 0.01 * torch.randn(N, 3, dtype=torch.float64)
    P["log_scales"][:] = -6.0
    D = to_dev(P, dev)
    acam = amd_cam(cam)
    sp = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 0)
    gb0, ts0, sg0, st0 = ops.bin_sort(acam, sp.clone(), ops.IntersectBudget(capacity=40 * N, sync=False))
    b = ops.IntersectBudget(capacity=40 * N, sync=False, max_list_hint=hint)
    sp1 = sp.clone()
    gb, ts, sg, st = ops.bin_sort(acam, sp1, b)
    n, ovf, need, longest = b.last_status4.tolist()
    lens = (ts0[1:] - ts0[:-1]).cpu().numpy()
    assert longest == lens.max() and n == int(st0.tolist()[0])
    assert torch.equal(ts, ts0)
    if ok:
        assert ovf == 0 and int(b.sticky.item()) == 0
        assert torch.equal(sg[:n], sg0[:n])
        assert b.check() == n
    else:
        assert ovf == 1 and int(b.sticky.item()) == 1
        a, r = sg[:n].cpu().numpy(), sg0[:n].cpu().numpy()
        starts = ts0.cpu().numpy()
        for t in np.nonzero(lens)[0]:          # every list is a permutation of the tile's ids; short lists are sorted
            x, y = a[starts[t]:starts[t + 1]], r[starts[t]:starts[t + 1]]
            assert np.array_equal(np.sort(x), np.sort(y))
            if lens[t] <= 1024:
                assert np.array_equal(x, y)
        # the compositing kernels run on the void frame without faulting
        rgb, depth, fT, _ = ops.rasterize_fwd(acam, sp1, sg, ts)
        ops.rasterize_bwd(acam, sp1, gb, sg, ts, rgb, depth, fT, v_rgb=torch.ones_like(rgb))
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="max_list_hint"):
            b.check()


def test_sort_fallback_global(dev):
    """> 16384 Gaussians in one tile exercises the global-memory sort fallback."""
    from touch_gs_amd import ops
    N = 17000
    P, cam = scene(N, 32, 32, 0, 81)
    P["means"][:] = torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64) + 0.01 * torch.randn(N, 3, dtype=torch.float64)
    P["log_scales"][:] = -6.0
    D = to_dev(P, dev)
    acam = amd_cam(cam)
    sp = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 0)
    gb, ts, sg, st = ops.bin_sort(acam, sp)
    f = splat_fields(sp)
    gid, tstart = O.bin_and_sort(f["rect"], f["hits"] > 0, f["depth"], cam)
    n = st.tolist()[0]
    assert np.diff(tstart).max() > 16384
    assert n == len(gid) and np.array_equal(sg[:n].cpu().numpy().astype(np.int64), gid)


def test_capacity_overflow_regrows(dev):
    from touch_gs_amd import ops
    P, cam = scene(4000, 160, 96, 0, 91)
    D, sp, _ = _project(dev, P, cam, 0)
    b = ops.IntersectBudget(capacity=100)
    gb, ts, sg, st = ops.bin_sort(amd_cam(cam), sp, b)
    n, ovf = st.tolist()
    assert ovf == 0 and n > 100 and b.capacity >= n
    b2 = ops.IntersectBudget(capacity=100, sync=False)
    ops.bin_sort(amd_cam(cam), sp, b2)
    with pytest.raises(RuntimeError):
        b2.check()


def test_adam_matches_torch(dev):
    import ctypes as C
    from touch_gs_amd import _lib
    from touch_gs_amd.optim import FusedAdam, GaussianParams
    N, K = 1000, 16
    torch.manual_seed(0)
    gp = GaussianParams.allocate(N, K, dev)
    gp.flat.copy_(torch.randn_like(gp.flat))
    lrs = dict(means=1.6e-4, log_scales=5e-3, quats=1e-3, opac_logit=5e-2, sh_dc=2.5e-3, sh_rest=1.25e-4)
    ref_p = [gp.means.clone().requires_grad_(True), gp.log_scales.clone().requires_grad_(True),
             gp.quats.clone().requires_grad_(True), gp.opac_logit.clone().requires_grad_(True),
             gp.sh[:, :1].clone().requires_grad_(True), gp.sh[:, 1:].clone().requires_grad_(True)]
    opt_ref = torch.optim.Adam([dict(params=[p], lr=lr) for p, lr in zip(ref_p, lrs.values())], eps=1e-15)
    opt = FusedAdam(gp, lrs, eps=1e-15)
    for step in range(3):
        g = torch.randn_like(gp.flat)
        gp.grad.copy_(g)
        gv = GaussianParams.views_of(g, N, K)
        for p, gg in zip(ref_p, [gv["means"], gv["log_scales"], gv["quats"], gv["opac_logit"], gv["sh"][:, :1], gv["sh"][:, 1:]]):
            p.grad = gg.clone()
        opt_ref.step()
        opt.step()
    got = [gp.means, gp.log_scales, gp.quats, gp.opac_logit, gp.sh[:, :1], gp.sh[:, 1:]]
    for a, b in zip(got, ref_p):
        assert torch.allclose(a, b.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("W,H", [(64, 48), (100, 70), (33, 17)])
def test_ssim_fwd_bwd(dev, W, H):
    from touch_gs_amd import ops
    g = torch.Generator().manual_seed(W)
    a = torch.rand(H, W, 3, generator=g, dtype=torch.float64)
    b = (a + 0.2 * torch.randn(H, W, 3, generator=g, dtype=torch.float64)).clamp(0, 1)
    a.requires_grad_(True)
    s = O.ssim(a, b)
    (0.2 * (1 - s)).backward()
    tot, v = ops.ssim_fwd_bwd(a.detach().float().to(dev), b.float().to(dev), weight=-0.2 / (3 * H * W))
    assert abs(tot.item() / (3 * H * W) - s.item()) < 1e-5
    scale = a.grad.abs().max().item()
    assert (v.cpu().double() - a.grad).abs().max().item() < 1e-4 * scale


_SSIM_WORKER = r'''
import sys, torch, ctypes as C
sys.path.insert(0, sys.argv[1])
from touch_gs_amd import _lib, ops
dev = torch.device("cuda:0")
out = {}
for W, H in ((54, 35), (55, 36), (200, 123), (640, 360), (1000, 57)):
    g = torch.Generator().manual_seed(W * 7 + H)
    a = torch.rand(H, W, 3, generator=g).to(dev)
    b = (a + 0.2 * torch.randn(H, W, 3, generator=g).to(dev)).clamp(0, 1)
    tot, v = ops.ssim_fwd_bwd(a, b, weight=-0.2 / (3 * H * W))
    out[f"{W}x{H}/v"] = v.cpu(); out[f"{W}x{H}/tot"] = tot.double().cpu()
    # bands (tgs_ssim_fwd_bwd_rows): v_img rows [y0, y1), the map summed over the same rows
    lib = _lib.load()
    nb = ((H + 15) // 16) * ((W + 15) // 16)
    vb = torch.zeros_like(a); scratch = torch.empty(9 * H * W, device=dev)
    tots = []
    for y0, y1 in ((0, H // 3), (H // 3, H // 3 + 7), (H // 3 + 7, H)):
        bp = torch.empty(nb, device=dev)
        _lib.check(lib.tgs_ssim_fwd_bwd_rows(W, H, _lib.ptr(a), _lib.ptr(b), C.c_float(-0.2 / (3 * H * W)), _lib.ptr(bp), nb,
                                             _lib.ptr(vb), _lib.ptr(scratch), y0, y1, y0, y1, None), "rows")
        tots.append(bp.double().sum())
    out[f"{W}x{H}/vb"] = vb.cpu(); out[f"{W}x{H}/totb"] = sum(tots).cpu()
torch.save(out, sys.argv[2])
'''


def test_ssim_one_pass_equals_two_kernels(dev, tmp_path):
    """k_ssim_fused (forward + backward of a strip in one pass, adjoint planes in LDS; the default) against the
    two-kernel path with the adjoint planes in HBM (TGS_SSIM_FUSED=0, read once per process: two worker processes):
    v_img bit for bit -- whole images and bands, sizes around the 54-column / segment boundaries -- and the map sum to
    fp32 summation order."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fused in ("1", "0"):
        f = str(tmp_path / f"ssim_{fused}.pt")
        env = dict(os.environ, TGS_SSIM_FUSED=fused)
        r = subprocess.run([sys.executable, "-c", _SSIM_WORKER, root, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[fused] = torch.load(f)
    for k, v in res["1"].items():
        if k.endswith("/v") or k.endswith("/vb"):
            assert torch.equal(v, res["0"][k]), (k, (v - res["0"][k]).abs().max())
        else:
            assert abs(float(v) - float(res["0"][k])) < 2e-6 * abs(float(res["0"][k])) + 1e-6, (k, float(v), float(res["0"][k]))
    for k in [k for k in res["1"] if k.endswith("/v")]:      # the bands together = the whole image
        assert torch.equal(res["1"][k], res["1"][k + "b"]), k


def test_selforacle_fixtures_gpu(dev):
    """HIP pipeline against the committed known-answer fixtures (tests/golden/raster_selforacle.npz)."""
    import os
    from touch_gs_amd import Camera, ops
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raster_selforacle.npz"))
    for name in ("one", "seven", "two_hundred"):
        N, W, H, deg, seed = [int(v) for v in z[f"{name}/meta"]]
        fx, fy, cx, cy = z[f"{name}/intr"]
        cam = Camera(z[f"{name}/viewmat"], fx, fy, cx, cy, W, H, bg=tuple(z[f"{name}/bg"]))
        D = {k: torch.from_numpy(z[f"{name}/in/{k}"]).float().to(dev).requires_grad_(True)
             for k in ("means", "log_scales", "quats", "opac_logit", "sh")}
        rgb, depth, alpha, _ = ops.render(D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], cam, deg)
        for got, k in ((rgb, "rgb"), (depth, "depth_acc"), (alpha, "alpha")):
            e = relerr(got.detach().cpu().numpy(), z[f"{name}/out/{k}"], floor=1e-2)
            assert np.quantile(e, 0.995) < TOL, (name, k, np.quantile(e, 0.995))
        w = lambda k: torch.from_numpy(z[f"{name}/{k}"]).float().to(dev)
        ((rgb * w("w_rgb")).sum() + (depth * w("w_depth")).sum() + (alpha * w("w_alpha")).sum()).backward()
        for k in D:
            ref = z[f"{name}/grad/{k}"]
            got = D[k].grad.cpu().double().numpy()
            cos = (ref * got).sum() / max(np.sqrt((ref * ref).sum() * (got * got).sum()), 1e-30)
            assert cos > 0.9999, (name, k, cos)


def test_fused_project_bin_sort_equals_separate_calls(dev):
    """tgs_project_bin_sort (K1 fused with the tile count) == tgs_project_fwd + tgs_bin_sort."""
    from touch_gs_amd import ops
    for N, W, H, deg, seed in ((5000, 200, 120, 3, 201), (777, 70, 50, 1, 202), (300, 64, 48, 0, 203)):
        P, cam = scene(N, W, H, deg, seed)
        acam = amd_cam(cam)
        D = to_dev(P, dev)
        sp1, r1 = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, want_radii=True)
        gb1, ts1, sg1, st1 = ops.bin_sort(acam, sp1)
        sp2, r2, gb2, ts2, sg2, st2 = ops.project_bin_sort(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                            D["sh"], deg, want_radii=True)
        n = int(ts1[-1])
        assert st2.tolist() == st1.tolist() and st1.tolist()[1] == 0 and torch.equal(r1, r2) and torch.equal(ts1, ts2)
        assert 0 < n <= st1.tolist()[0] and torch.equal(sg1[:n], sg2[:n])
        assert torch.equal(sp1.view(torch.int32), sp2.view(torch.int32))  # records incl. rect + in-group offset


@pytest.mark.parametrize("active_deg", [0, 1, 2])
def test_partial_sh_degree_with_full_storage(dev, active_deg):
    """SH tensor stored at degree 3 (K = 16) but evaluated at a lower active degree, as during
    the sh_degree_interval warm-up: forward colours and all gradients vs the oracle; the unused
    coefficient gradients must be exactly zero."""
    from touch_gs_amd import ops
    N, W, H = 2500, 144, 96
    P, cam = scene(N, W, H, 3, 300 + active_deg)
    D = {k: v.requires_grad_(True) for k, v in to_dev(P, dev).items()}
    rgb, depth, alpha, _ = ops.render(D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"],
                                      amd_cam(cam), active_deg)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    out, *_ = O.render(Pg["means"], Pg["log_scales"], Pg["quats"], Pg["opac_logit"], Pg["sh"], cam, active_deg)
    e = relerr(rgb.detach().cpu().numpy(), out["rgb"].detach().numpy(), floor=1e-2)
    assert np.quantile(e, 0.99) < TOL
    g = torch.Generator().manual_seed(9)
    w = torch.randn(H, W, 3, generator=g, dtype=torch.float64)
    wd = torch.randn(H, W, generator=g, dtype=torch.float64)
    ((out["rgb"] * w).sum() + (out["depth_acc"] * wd).sum()).backward()
    ((rgb * w.float().to(dev)).sum() + (depth * wd.float().to(dev)).sum()).backward()
    K = (active_deg + 1) ** 2
    assert float(D["sh"].grad[:, K:].abs().max()) == 0.0
    for k in D:
        ref, got = Pg[k].grad.numpy(), D[k].grad.cpu().double().numpy()
        cos = (ref * got).sum() / np.sqrt((ref * ref).sum() * (got * got).sum())
        assert cos > 0.9999, (k, cos)


def test_degenerate_sizes(dev):
    from touch_gs_amd import Camera, ops
    # N = 0: background only
    cam = Camera(np.eye(4), 50.0, 50.0, 8.0, 8.0, 16, 16, bg=(0.5, 0.25, 0.125))
    z = lambda *s: torch.zeros(*s, device=dev)
    sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, z(0, 3), z(0, 3), z(0, 4), z(0), z(0, 16, 3), 3)
    assert st.tolist() == [0, 0]
    rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    assert torch.allclose(rgb, torch.tensor([0.5, 0.25, 0.125], device=dev).expand(16, 16, 3)) and float(fT.min()) == 1.0
    # a 1x1 and a 17x5 image (partial tiles in both directions), one Gaussian in front of the camera
    for W, H in ((1, 1), (17, 5)):
        cam = Camera(np.eye(4), 20.0, 20.0, W / 2, H / 2, W, H)
        means = torch.tensor([[0.0, 0.0, 2.0]], device=dev)
        ls = torch.full((1, 3), -1.0, device=dev)
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
        op = torch.tensor([2.0], device=dev)
        sh = torch.zeros(1, 1, 3, device=dev)
        sh[0, 0] = torch.tensor([1.0, 0.0, -1.0])
        for t in (means, ls, q, op, sh):
            t.requires_grad_(True)
        rgb, depth, alpha, radii = ops.render(means, ls, q, op, sh, cam, 0)
        assert rgb.shape == (H, W, 3) and int(radii[0]) > 0 and float(alpha.max()) > 0.5
        (rgb.sum() + depth.sum()).backward()
        assert all(torch.isfinite(t.grad).all() for t in (means, ls, q, op, sh))
        ocam = O.Camera(viewmat=torch.eye(4, dtype=torch.float64), fx=20.0, fy=20.0, cx=W / 2, cy=H / 2, W=W, H=H)
        o, *_ = O.render(means.detach().cpu().double(), ls.detach().cpu().double(), q.detach().cpu().double(),
                         op.detach().cpu().double(), sh.detach().cpu().double(), ocam, 0)
        assert np.abs(rgb.detach().cpu().numpy() - o["rgb"].numpy()).max() < 1e-5


@pytest.mark.parametrize("W,H", [(208, 144), (70, 50), (16, 16), (1920, 1080), (3840, 2160)])
def test_tile_order_schedule(dev, W, H):
    """tgs_bin_sort's tile_order: block b (XCD b % 8) gets a tile of every 8th 8-tile granule, every chunk
    of <= 1024 slots of an XCD is visited longest list first (ties by tile id; 3840x2160 has four chunks per
    XCD), every tile appears exactly once, padding entries are T; the compositing kernels give
    bit-identical results with and without the schedule."""
    from touch_gs_amd import ops
    P, cam = scene(3000, W, H, 2, 23)
    acam = amd_cam(cam)
    D = to_dev(P, dev)
    splats = ops.project_fwd(acam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], 2)
    gb, ts, sg, st = ops.bin_sort(acam, splats)
    T = acam.num_tiles
    TW, TH = acam.tiles
    order = ts.tile_order.cpu().numpy()
    n = np.diff(ts.cpu().numpy())
    G = 8                                        # granule: XCD x owns the tiles t with (t // 8) % 8 == x
    per = -(-(-(-T // G)) // 8) * G              # slots per XCD
    assert order.shape[0] == per * 8
    assert sorted(order[order < T].tolist()) == list(range(T)) and int((order == T).sum()) == per * 8 - T
    chunk = -(-(-(-per // -(-per // 1024))) // G) * G   # <= 1024 slots, a multiple of the granule
    n_sub = -(-per // chunk)
    assert n_sub == ops._lib.load().tgs_num_bands(W, H)
    for x in range(8):
        mine = order[x::8]
        assert np.all((mine[mine < T] // G) % 8 == x)
        for c in range(n_sub):                   # every chunk of <= 1024 slots: its own tiles, longest list first
            got = mine[c * chunk:(c + 1) * chunk]
            slots = np.arange(c * chunk, min((c + 1) * chunk, per))
            want = ((slots // G) * 8 + x) * G + slots % G
            want = want[want < T]
            got = got[got < T]
            assert sorted(got.tolist()) == sorted(want.tolist())
            keys = [(-int(n[t]), int(t)) for t in got]
            assert keys == sorted(keys)
    out_a = ops.rasterize_fwd(acam, splats, sg, ts)
    bwd_a = ops.rasterize_bwd(acam, splats, gb, sg, ts, out_a[0], out_a[1], out_a[2],
                              v_rgb=torch.ones_like(out_a[0]), v_depth=torch.ones_like(out_a[1]))[0]
    del ts.tile_order                      # wrappers now pass NULL: spatial order
    out_b = ops.rasterize_fwd(acam, splats, sg, ts)
    bwd_b = ops.rasterize_bwd(acam, splats, gb, sg, ts, out_b[0], out_b[1], out_b[2],
                              v_rgb=torch.ones_like(out_b[0]), v_depth=torch.ones_like(out_b[1]))[0]
    for a, b in zip(out_a[:3], out_b[:3]):
        assert torch.equal(a, b)
    # the pair index space has holes (one region per XCD): compare the slots that exist, through K8a
    assert torch.equal(ops.reduce_partials(acam, splats, gb, bwd_a), ops.reduce_partials(acam, splats, gb, bwd_b))


def test_long_run_threshold_changes_nothing_but_the_shape_of_the_sums(dev):
    """tgs_set_long_run (which Gaussians count as long runs: outside the group's counting box, their partial records
    summed by the whole workgroup in K8) is a launch-shape parameter: on an object-centric scene with screen-filling
    Gaussians the tile lists are the same bit for bit at 4, 8, 32 and 256 tiles, the images too, and the parameter
    gradients agree to the rounding of K8's sums."""
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    N, W, H, deg = 40_000, 640, 400, 3
    P, intr = synthetic_gaussians(N, W, H, deg, 9, clustered=True)
    P["log_scales"][::400] += 3.5                       # a hundred Gaussians covering hundreds of tiles
    D = {k: v.to(dev).float().contiguous() for k, v in P.items()}
    cam = make_camera(intr, 1, 8, bg=(0.1, 0.2, 0.3))
    g = torch.Generator().manual_seed(4)
    v_rgb = torch.randn(H, W, 3, generator=g).to(dev)
    v_d, v_a = torch.randn(H, W, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev)
    before = ops.set_long_run()
    out = {}
    try:
        for lr in (32, 4, 8, 256):
            assert ops.set_long_run(lr) == lr
            sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg)
            T = cam.num_tiles
            n = int(ts[T])
            rgb, depth, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
            partials, _ = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, v_rgb, v_d, v_a)
            grads = ops.project_bwd(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], deg, sp, gb, partials)
            out[lr] = (ts[:T + 1].clone(), sg[:n].clone(), rgb.clone(), depth.clone(), [t.clone() for t in grads[:5]])
    finally:
        ops.set_long_run(before)
    hits = (out[32][0][1:] - out[32][0][:-1])
    assert int(hits.max()) > 64
    for lr in (4, 8, 256):
        a, b = out[32], out[lr]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), lr            # the lists
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), lr            # the images
        for x, y in zip(a[4], b[4]):
            assert torch.allclose(x, y, rtol=2e-4, atol=1e-6 * float(x.abs().max())), (lr, float((x - y).abs().max()), float(x.abs().max()))
    assert any(not torch.equal(x, y) for x, y in zip(out[32][4], out[4][4]))       # (the sums did take another shape)

