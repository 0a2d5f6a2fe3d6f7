"""Scratch timing of each kernel on the BASELINE configs (developer tool, not bench.py)."""
import sys, time, math
import torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from oracle import torch_oracle as O
from touch_gs_amd import ops, Camera

def morton_order(means, bits=10):
    m = means.double()
    lo, hi = m.min(0).values, m.max(0).values
    q = ((m - lo) / (hi - lo + 1e-12) * (2 ** bits - 1)).long()
    code = torch.zeros(len(m), dtype=torch.long)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code)


def run(N, W, H, deg, seed, iters=10, morton=False):
    dev = torch.device('cuda:0')
    P, c = O.synthetic_scene(N, W, H, deg, seed, dtype=torch.float32)
    if morton:
        order = morton_order(P['means'])
        P = {k: v[order] for k, v in P.items()}
    D = {k: v.to(dev).contiguous() for k, v in P.items()}
    cam = Camera(O.orbit_viewmat(0, 8).numpy(), c['fx'], c['fy'], c['cx'], c['cy'], W, H)
    budget = ops.IntersectBudget()
    sp = ops.project_fwd(cam, D['means'], D['log_scales'], D['quats'], D['opac_logit'], D['sh'], deg)
    gb, ts, sg, st = ops.bin_sort(cam, sp, budget)
    n = st.tolist()[0]
    print(f'morton={morton} N={N} {W}x{H} I={n} tiles={cam.num_tiles} per-tile={n/cam.num_tiles:.0f} max-tile={int((ts[1:]-ts[:-1]).max())} visible={int((sp[:,4]>0).sum())}')
    budget.sync = False
    budget.capacity = int(n * 1.2)
    gt = torch.rand(H, W, 3, device=dev); dgt = torch.rand(H, W, device=dev) * 5; unc = torch.rand(H, W, device=dev)
    spec = dict(gt_rgb=gt, gt_depth=dgt, uncertainty=unc, l1_weight=0.8/(3*H*W), depth_weight=0.2/(H*W), uncertainty_weight=1.0)
    def T(fn, name):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): r = fn()
        e1.record(); torch.cuda.synchronize()
        print(f'  {name:16s} {e0.elapsed_time(e1)/iters*1e3:9.1f} us')
        return r
    sp = T(lambda: ops.project_fwd(cam, D['means'], D['log_scales'], D['quats'], D['opac_logit'], D['sh'], deg), 'project_fwd')
    gb, ts, sg, st = T(lambda: ops.bin_sort(cam, sp, budget), 'bin_sort')
    rgb, depth, fT, fidx = T(lambda: ops.rasterize_fwd(cam, sp, sg, ts), 'raster_fwd')
    fidx = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=True)[3]
    partials, tl = T(lambda: ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, loss=spec, want_tile_loss=True), 'raster_bwd')
    T(lambda: ops.project_bwd(cam, D['means'], D['log_scales'], D['quats'], D['opac_logit'], D['sh'], deg, sp, gb, partials), 'project_bwd')
    T(lambda: ops.ssim_fwd_bwd(rgb, gt, -0.2/(3*H*W)), 'ssim')
    print('  alpha mean', float((1-fT).mean()), 'mean last idx', float(fidx.float().mean()))

if __name__ == '__main__':
    run(1_000_000, 1920, 1080, 3, 1236)
    run(1_000_000, 1920, 1080, 3, 1236, morton=True)
