"""Developer probe: K6 / K7 time with the library's tile schedule (XCD x = horizontal image band x)
against a schedule that interleaves tile rows over the XCDs (row r -> XCD r % 8), per view."""
import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, _ = synthetic_gaussians(N, W, H, deg, 1236)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), params)
model.spatial_sort()
p = model.params
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1

def interleaved(ts, TW, TH):
    T = TW * TH
    n = (ts[1:] - ts[:-1]).long()
    tile = torch.arange(T, device=dev)
    x = ((tile // TW) // S) % 8
    per = (T + 7) // 8
    # longest first inside each XCD; surplus tiles of an XCD (rows do not divide evenly) go to the emptiest
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

for vi in range(8):
    view = make_view(N, W, H, deg, 1236, dev, view=vi, n_views=8)
    sp, _, gb, ts, sg, st = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
    rgb, dacc, fT, _ = ops.rasterize_fwd(view.cam, sp, sg, ts)
    _, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    k6 = lambda: ops.rasterize_fwd(view.cam, sp, sg, ts)
    k7 = lambda: ops.rasterize_bwd(view.cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=v_img, loss=model.loss_spec(view), want_tile_loss=True)
    lib_order = ts.tile_order
    a6, a7 = timed(k6), timed(k7)
    ts.tile_order = interleaved(ts, *view.cam.tiles)
    assert sorted(ts.tile_order[ts.tile_order < view.cam.num_tiles].tolist()) == list(range(view.cam.num_tiles))
    b6, b7 = timed(k6), timed(k7)
    ts.tile_order = lib_order
    print(f"view {vi}: pairs {int(ts[-1])}  K6 {a6:.0f} -> {b6:.0f} us   K7 {a7:.0f} -> {b7:.0f} us", flush=True)
