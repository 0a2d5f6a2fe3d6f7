"""Developer tool: run K1..K8 a few times on cfg3 (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, '..')
from touch_gs_amd import ops
from touch_gs_amd.scene import synthetic_gaussians, make_camera
N, W, H, deg = 1_000_000, 1920, 1080, 3
dev = torch.device('cuda:0')
P, intr = synthetic_gaussians(N, W, H, deg, 1236)
D = {k: v.to(dev).contiguous() for k, v in P.items()}
cam = make_camera(intr, 0, 8)
gt = torch.rand(H, W, 3, device=dev); dgt = torch.rand(H, W, device=dev) * 5; unc = torch.rand(H, W, device=dev)
spec = dict(gt_rgb=gt, gt_depth=dgt, uncertainty=unc, l1_weight=0.8/(3*H*W), depth_weight=0.2/(H*W), uncertainty_weight=1.0)
b = ops.IntersectBudget()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    sp = ops.project_fwd(cam, D['means'], D['log_scales'], D['quats'], D['opac_logit'], D['sh'], deg)
    gb, ts, sg, st = ops.bin_sort(cam, sp, b)
    rgb, depth, fT, fidx = ops.rasterize_fwd(cam, sp, sg, ts)
    partials, tl = ops.rasterize_bwd(cam, sp, gb, sg, ts, rgb, depth, fT, loss=spec, want_tile_loss=True)
    ops.project_bwd(cam, D['means'], D['log_scales'], D['quats'], D['opac_logit'], D['sh'], deg, sp, gb, partials)
torch.cuda.synchronize()
