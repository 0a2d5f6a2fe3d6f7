"""Developer tool: single-GPU timing of the kernels of the data-parallel step on cfg3 (K8 writing the
colour-gradient block, SH Adam from `world` gathered blocks, geometry Adam) next to the dense ones."""
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
view = make_view(N, W, H, deg, 1236, dev, view=0, n_views=8)
view.valid_count()
opt, p = model.optimizer, model.params
block = torch.zeros(3 * N + 4, device=dev)
model.forward_backward(view, color_block=block)
L = model.last


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


# rebuild the K8 inputs once
cam = view.cam
splats, radii, group_base, tile_start, sorted_gid, _ = ops.project_bin_sort(
    cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, model.budget)
rgb, depth_acc, fT, _ = ops.rasterize_fwd(cam, splats, sorted_gid, tile_start)
partials, _ = ops.rasterize_bwd(cam, splats, group_base, sorted_gid, tile_start, rgb, depth_acc, fT,
                                loss=model.loss_spec(view), want_tile_loss=True)
print("K8 dense   us", timeit(lambda: ops.project_bwd(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, splats,
                                                      group_base, partials, out=p.grad_views())))
print("K8 colour  us", timeit(lambda: ops.project_bwd_color(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg,
                                                            splats, group_base, partials, p.grad_views()[:4], block)))
opt.begin_step()
print("Adam dense us", timeit(lambda: opt.step_range(0, -1, 1.0)))
ge = opt.geom_end()
print("Adam geom  us", timeit(lambda: opt.step_range(0, ge, 1.0)))
for world in (1, 2, 4, 8):
    allc = block.repeat(world).contiguous()
    print(f"Adam SH gathered world={world} us", timeit(lambda: opt.step_sh_gathered(world, deg, allc, 1.0 / world)))
