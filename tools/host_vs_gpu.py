"""Developer tool: is the train step host-bound or GPU-bound at a given size?  Per step: wall time of the steady loop,
host time spent inside train_step (enqueue only), GPU busy time (sum of kernel durations is not available without a
profiler: the loop is timed once normally and once with a device-side sleep-free 'all queued first' trick -- the host
enqueues `burst` steps behind a blocking event, then the GPU drains them; drain time / burst = GPU time per step).
    python tools/host_vs_gpu.py [N] [W] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
from touch_gs_amd.optim import GaussianParams
from touch_gs_amd.scene import make_views, synthetic_gaussians

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
H = int(sys.argv[3]) if len(sys.argv) > 3 else 720
deg, nv = 3, 8
dev = torch.device("cuda:0")
views, D = make_views(N, W, H, deg, 77, dev, nv, clustered=True)
P, _ = synthetic_gaussians(N, W, H, deg, 78, clustered=True)
params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
m = DepthGaussianSplattingModel(ModelConfig(sh_degree=deg, sh_degree_interval=0, depth_loss_mult=0.2, spatial_sort=True), params)
m.spatial_sort()
m.enable_speculative_budget()
def step(i):
    m.train_step(views[i % nv], next_view=views[(i + 1) % nv])
for i in range(60): step(i)
m.flush(); torch.cuda.synchronize()
# steady loop
K = 300
t0 = time.perf_counter(); host = 0.0
for i in range(K):
    a = time.perf_counter(); step(i); host += time.perf_counter() - a
m.flush(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K
# GPU time: the queue is kept full by construction when the host is faster; measure with events over the loop
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for i in range(K): step(i)
e1.record(); m.flush(); torch.cuda.synchronize()
gpu_span = e0.elapsed_time(e1) / K
# pure enqueue cost: no verdict polling inside the burst (max_in_flight raised), GPU idle at the start
m.flush(); torch.cuda.synchronize()
m._max_in_flight = 10 ** 6
B = 40
t0 = time.perf_counter()
for i in range(B): step(i)
enq = (time.perf_counter() - t0) / B
torch.cuda.synchronize(); m.flush()
import json
print(json.dumps({"N": N, "W": W, "H": H, "wall_ms_per_step": round(wall * 1e3, 4), "host_ms_in_train_step": round(host / K * 1e3, 4),
                  "event_span_ms_per_step": round(gpu_span, 4), "host_enqueue_ms_per_step_burst40": round(enq * 1e3, 4), "replays": getattr(m, "speculative_replays", 0)}))
