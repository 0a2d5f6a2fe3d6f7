"""Same-box comparison of N builds of libtgs_hip.so (tools/build_variant.sh), alternating:

    python tools/abn.py [--cfg cfg3] [--rounds 3] [--views 8] [--morton] lib1.so lib2.so ...

Every round runs each library in a fresh subprocess (TGS_LIB_PATH) and prints, per kernel, the median
HIP-event time over 20 repetitions of every view, averaged over the views (plus view 0 alone), and a
checksum of the gradients so that variants which are meant to be result-preserving can be told apart
from ones that are not."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys, torch
sys.path.insert(0, %(root)r)
from touch_gs_amd import ops
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_view, synthetic_gaussians
cfg = {"cfg2": (100_000, 800, 800, 1235, False), "cfg3": (1_000_000, 1920, 1080, 1236, False),
       "cfg5": (5_000_000, 3840, 2160, 1238, False), "clustered": (1_000_000, 1920, 1080, 1236, True)}[%(cfg)r]
N, W, H, seed, cl = cfg
deg, nv = 3, %(views)d
dev = torch.device("cuda:0")
P, _ = synthetic_gaussians(N, W, H, deg, seed, clustered=cl)
p = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
model = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0), p)
if %(morton)d:
    model.spatial_sort()
    p = model.params
views = [make_view(N, W, H, deg, seed, dev, view=v, n_views=8, clustered=cl) for v in range(nv)]
need = longest = 0
for v in views:
    v.valid_count()
    b = ops.IntersectBudget()
    ops.project_bin_sort(v.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, b)
    need = max(need, b.last_need)
    longest = max(longest, b.last_longest)
budget = ops.IntersectBudget(capacity=int(need * 1.25) + 4096, sync=False,
                             max_list_hint=-1 if os.environ.get("TGS_AB_NO_LIST_HINT") else int(1.25 * longest) + 32)
ev = lambda: torch.cuda.Event(enable_timing=True)
names = ["front", "k6", "ssim", "k7", "k8"]
per_view = []
chk = 0.0
for vi, view in enumerate(views):
    evs = []
    for it in range(28):
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
    torch.cuda.synchronize()
    acc = {k: [] for k in names}
    for e in evs[8:]:
        for j, k in enumerate(names):
            acc[k].append(e[j].elapsed_time(e[j + 1]))
    per_view.append({k: sorted(v)[len(v) // 2] * 1e3 for k, v in acc.items()})
    chk += float(p.grad.double().abs().sum()) + float(rgb.double().sum())
budget.check()
out = {k: round(sum(d[k] for d in per_view) / len(per_view), 1) for k in names}
out["k6_v0"] = round(per_view[0]["k6"], 1); out["k7_v0"] = round(per_view[0]["k7"], 1)
out["chk"] = "%%.10e" %% chk
print(json.dumps(out))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--cfg", default="cfg3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--morton", action="store_true")
    a = ap.parse_args()
    code = WORKER % dict(root=ROOT, cfg=a.cfg, views=a.views, morton=int(a.morton))
    for r in range(a.rounds):
        for spec in a.libs:     # "lib.so" or "lib.so@ENV=VAL,ENV2=VAL2" (environment switches of the same build)
            lib, _, envs = spec.partition("@")
            env = dict(os.environ, TGS_LIB_PATH=os.path.abspath(lib))
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print(os.path.basename(spec), line[-1] if line else out.stderr[-800:], flush=True)


if __name__ == "__main__":
    main()
