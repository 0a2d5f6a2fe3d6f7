"""Same-box A/B timing of two builds of libtgs_hip.so (kernel times drift ~10 % between boxes, so
variants are only comparable when they alternate on ONE box after warm-up).

    python tools/ab.py A.so B.so [cfg3|cfg2|clustered] [rounds]

Every round runs each library in a fresh subprocess (TGS_LIB_PATH) and prints the median HIP-event
time of the front half, K6, SSIM, K7 and K8 over 30 repetitions."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, sys, torch
sys.path.insert(0, %(root)r)
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
cfg = {"cfg2": (100_000, 800, 800, 1235, False), "cfg3": (1_000_000, 1920, 1080, 1236, False),
       "clustered": (1_000_000, 1920, 1080, 1236, True)}[%(cfg)r]
N, W, H, seed, cl = cfg
deg = 3
dev = torch.device("cuda:0")
P, _ = synthetic_gaussians(N, W, H, deg, seed, clustered=cl)
p = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), p)
import os
if os.environ.get("TGS_AB_MORTON"):
    model.spatial_sort()
    p = model.params
view = make_view(N, W, H, deg, seed, dev, clustered=cl)
view.valid_count()
b = ops.IntersectBudget()
ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, b)
budget = ops.IntersectBudget(capacity=int(b.last_need * 1.25) + 4096, sync=False)
ev = lambda: torch.cuda.Event(enable_timing=True)
names = ["front", "k6", "ssim", "k7", "k8"]
acc = {k: [] for k in names}
evs = []
for it in range(45):
    e = [ev() for _ in range(6)]
    e[0].record()
    sp, _, gb, ts, sg, _ = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, budget)
    e[1].record()
    rgb, dacc, fT, _ = ops.rasterize_fwd(view.cam, sp, sg, ts)
    e[2].record()
    _, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-0.2 / (3 * H * W), reduce=False)
    e[3].record()
    partials, tl = ops.rasterize_bwd(view.cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=v_img, loss=model.loss_spec(view), want_tile_loss=True)
    e[4].record()
    ops.project_bwd(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, sp, gb, partials, out=p.grad_views())
    e[5].record()
    evs.append(e)
torch.cuda.synchronize()      # one synchronisation: no kernel starts on an idle GPU behind a host round trip
for e in evs[15:]:
    for j, k in enumerate(names):
        acc[k].append(e[j].elapsed_time(e[j + 1]))
print(json.dumps({k: round(sorted(v)[len(v) // 2] * 1e3, 1) for k, v in acc.items()}))
'''


def main():
    a, b = sys.argv[1], sys.argv[2]
    cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    code = WORKER % dict(root=ROOT, cfg=cfg)
    for r in range(rounds):
        for tag, lib in (("A", a), ("B", b)):
            env = dict(os.environ, TGS_LIB_PATH=os.path.abspath(lib))
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(tag, os.path.basename(lib), line[-1] if line else out.stderr[-500:], flush=True)


if __name__ == "__main__":
    main()
