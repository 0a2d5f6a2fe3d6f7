"""Developer tool: where does a tile's time go in K7's four-wave form?  Needs the instrumented build
(hipcc ... -DTGS_QUAD_TIMING raster.hip -> build_ab/qt.so; TGS_LIB_PATH=build_ab/qt.so).  Per schedule slot and wave (quadrant) the
kernel sums s_memtime ticks spent (a) at the batch's top barrier, (b) staging / waiting for the staging wave, (c) in its own walk,
(d) waiting for the slowest quadrant.  Prints the deepest tiles and the totals."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from touch_gs_amd import ops, _lib
import ckpt_loop
what = sys.argv[1] if len(sys.argv) > 1 else "bunny"
dev = torch.device("cuda:0")
m, views = ckpt_loop.load(os.path.join(ROOT, "build_ab/ckpt/model_%s_1.pt" % ("bunny_real" if what == "bunny" else "block")), dev, n_views=2)
lib = _lib.load()
lib.tgs_debug_quad_timing.restype = C.c_int
lib.tgs_debug_quad_timing.argtypes = [C.c_void_p, C.c_int]
p, deg = m.params, 3
v = views[0]; cam = v.cam; H, W = cam.H, cam.W; T = cam.num_tiles; TW, TH = cam.tiles
buf = np.zeros(8192 * 16, dtype=np.uint64)
for rep in range(3):
    sp, _, gb, ts, sg, st = ops.project_bin_sort(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg)
    rgb, dacc, fT, _ = ops.rasterize_fwd(cam, sp, sg, ts)
    ss, vimg = ops.ssim_fwd_bwd(rgb, v.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    torch.cuda.synchronize()
    lib.tgs_debug_quad_timing(None, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=vimg, loss=m.loss_spec(v), want_tile_loss=True); b.record()
    torch.cuda.synchronize()
    k7 = a.elapsed_time(b) * 1e3
assert lib.tgs_debug_quad_timing(buf.ctypes.data, 0) == 0
d = buf.reshape(8192, 4, 4).astype(np.float64)
n = (ts[1:T + 1] - ts[:T]).long()
pad = torch.zeros(TH * 16, TW * 16, dtype=torch.int64, device=dev)
pad[:H, :W] = fT.stop_pos.long().clamp(max=int(n.max()))
walk = torch.minimum(pad.view(TH, 16, TW, 16).permute(0, 2, 1, 3).reshape(T, 256).max(1).values, n).cpu().numpy()
order = ts.tile_order[:T].cpu().numpy() if hasattr(ts, "tile_order") else np.arange(T)
tot = d.sum(axis=2)                       # [slot][wave] ticks
used = np.nonzero(tot[:, 0] > 0)[0]
# tick length: the longest tile cannot take longer than the launch
tick_us = None
print(f"K7 {k7:.1f} us; slots with a four-wave tile: {len(used)}; longest tile total {tot[:, 0].max():.0f} ticks")
top = used[np.argsort(-tot[used, 0])][:12]
print("slot  tile  walk   ticks/wave0   conv-barriers  staging-next  own-walk  wait-slowest   (shares of wave 0 | of the slowest-walking wave)")
for s_ in top:
    t = int(order[s_]) if s_ < len(order) else -1
    w0 = d[s_, 0] / tot[s_, 0]
    kk = int(np.argmax(d[s_, :, 2]))
    wk = d[s_, kk] / tot[s_, kk]
    print(f"{s_:5d} {t:5d} {int(walk[t]) if 0 <= t < T else -1:5d} {tot[s_, 0]:10.0f}   " + " ".join(f"{x:6.2f}" for x in w0) + "   | k=%d " % kk + " ".join(f"{x:6.2f}" for x in wk))
S = d[used].sum(axis=(0, 1))
print("all four-wave tiles, all waves: conversion barriers %.2f staging the next batch %.2f own-walk %.2f wait-slowest %.2f" % tuple(S / S.sum()))
S0 = d[used][:, 0].sum(axis=0); print("wave 0 only: " + " ".join(f"{x:.2f}" for x in S0 / S0.sum()))
print(json.dumps(dict(k7_us=k7, ticks_longest=float(tot[:, 0].max()), walk_max=int(walk.max()))))
