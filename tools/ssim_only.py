import sys, torch
sys.path.insert(0, '.')
from touch_gs_amd import ops
H, W = 1080, 1920
dev = torch.device('cuda:0')
a = torch.rand(H, W, 3, device=dev); b = torch.rand(H, W, 3, device=dev)
for _ in range(5):
    ops.ssim_fwd_bwd(a, b, -0.2 / (3 * H * W))
torch.cuda.synchronize()
