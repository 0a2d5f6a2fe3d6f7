"""Developer probe: K6 / K7 time with the library's tile schedule against a hand-made one (here: 2-D
granules of BW x BH tiles dealt to the XCDs), per orbit view.  The first version of this probe compared
round 2a's schedule (one horizontal image band per XCD) with row interleaving and led to section 5.5."""
import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
import os
N, W, H, deg = (5_000_000, 3840, 2160, 3) if os.environ.get('PROBE_CFG') == 'cfg5' else (1_000_000, 1920, 1080, 3)
SEED = 1238 if N > 1_000_000 else 1236
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, SEED)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
model.spatial_sort()
p = model.params
BW, BH = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 2)   # 2-D granule (tiles)

def interleaved(ts, TW, TH):
    """XCD x owns the BW x BH tile blocks q (row major over the blocks) with q % 8 == x, longest list first."""
    T = TW * TH
    n = (ts[1:] - ts[:-1]).long()
    tile = torch.arange(T, device=dev)
    q = ((tile // TW) // BH) * ((TW + BW - 1) // BW) + (tile % TW) // BW
    x = q % 8
    per = ts.tile_order.numel() // 8
    key = x * (1 << 40) + ((1 << 20) - n) * (1 << 20) + tile
    srt = tile[torch.argsort(key)]
    xs = x[srt]
    lists = [srt[xs == k] for k in range(8)]
    extra = torch.cat([l[per:] for l in lists])
    lists = [l[:per] for l in lists]
    out = torch.full((per, 8), T, dtype=torch.int32, device=dev)
    e = 0
    for k in range(8):
        l = lists[k]
        if len(l) < per:
            take = min(per - len(l), len(extra) - e)
            l = torch.cat([l, extra[e:e + take]]); e += take
        out[:len(l), k] = l.int()
    return out.reshape(-1).contiguous()

def timed(f, reps=12):
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs[2:])
    return t[len(t) // 2] * 1e3

for vi in (range(8) if N <= 1_000_000 else (0, 3)):
    view = make_view(N, W, H, deg, SEED, dev, view=vi, n_views=8)
    sp, _, gb, ts, sg, st = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
    rgb, dacc, fT, _ = ops.rasterize_fwd(view.cam, sp, sg, ts)
    _, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    k6 = lambda: ops.rasterize_fwd(view.cam, sp, sg, ts)
    k7 = lambda: ops.rasterize_bwd(view.cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=v_img, loss=model.loss_spec(view), want_tile_loss=True)
    lib_order = ts.tile_order
    mine = interleaved(ts, *view.cam.tiles)
    assert sorted(mine[mine < view.cam.num_tiles].tolist()) == list(range(view.cam.num_tiles))
    same = bool(torch.equal(mine, lib_order))
    timed(k6, 4); timed(k7, 4)                      # warm-up
    ts.tile_order = mine
    b6, b7 = timed(k6), timed(k7)
    ts.tile_order = lib_order
    a6, a7 = timed(k6), timed(k7)                   # the library's schedule is timed SECOND
    print(f"view {vi}: pairs {int(ts[-1])}  K6 {a6:.0f} -> {b6:.0f} us   K7 {a7:.0f} -> {b7:.0f} us   identical tables: {same}", flush=True)
