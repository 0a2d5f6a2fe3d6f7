"""SSIM forward + backward at 1080p (and 720p / 800x800): the one-pass kernel (TGS_SSIM_FUSED=1, default) against the two-kernel
path, over TGS_SSIM_NBLK; both switches are read once per process, so every point is a fresh subprocess.
   python tools/ssim_sweep.py            (median of 30 calls, HIP events, us)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, torch
sys.path.insert(0, %r)
from touch_gs_amd import ops
dev = torch.device("cuda:0")
res = {}
for W, H in ((1920, 1080), (1280, 720), (800, 800), (3840, 2160)):
    a = torch.rand(H, W, 3, device=dev); b = torch.rand(H, W, 3, device=dev)
    for _ in range(5):
        ops.ssim_fwd_bwd(a, b, -0.2 / (3 * H * W), reduce=False)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.ssim_fwd_bwd(a, b, -0.2 / (3 * H * W), reduce=False); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    res["%%dx%%d" %% (W, H)] = round(sorted(ts)[len(ts) // 2], 1)
print(__import__("json").dumps(res))
''' % ROOT
for fused, nblks in (("0", ("",)), ("1", ("", "4", "5", "6", "7", "8", "9"))):
    for nb in nblks:
        env = dict(os.environ, TGS_SSIM_FUSED=fused)
        if nb:
            env["TGS_SSIM_NBLK"] = nb
        r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"fused={fused} nblk={nb or 'auto'}", line[-1] if line else r.stderr[-500:], flush=True)
