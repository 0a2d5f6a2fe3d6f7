#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its config: train iters/s (+ render Mpix/s) of the full
Touch-GS train step, 1 M synthetic Gaussians @ 1920x1080, SH degree 3, on N MI355X.

One "step" per rank = project + bin/sort + RGB&depth compositing + SSIM + compositing backward with
the fused L1 + tactile depth/uncertainty loss + projection backward + (N>1: RCCL all-reduce of the
flat gradient buffer) + fused Adam.  Inputs are resident in HBM before the timed region.
`value` = optimizer iterations/s x ranks = views/s of the whole job (weak scaling: one view per
rank per iteration).

    python bench.py [--gpus N --steps K --warmup W]        (N > 1 without WORLD_SIZE: launches its own N ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); ~6290 measured copy ceiling


CONFIGS = {  # BASELINE.json configs (SURVEY 8(d) seeds) + the clustered stress scene of DESIGN.md
    "cfg2": dict(gaussians=100_000, width=800, height=800, seed=1235,
                 label="configs[1]: 100k synthetic Gaussians, 800x800"),
    "cfg3": dict(gaussians=1_000_000, width=1920, height=1080, seed=1236,
                 label="configs[2]: 1M Gaussians, 1080p"),
    "cfg5": dict(gaussians=5_000_000, width=3840, height=2160, seed=1238,
                 label="configs[4] (per-GPU part): 5M Gaussians, SH degree 3, 4K"),
    "clustered": dict(gaussians=1_000_000, width=1920, height=1080, seed=1236, clustered=True,
                      label="stress: 1M Gaussians, 1080p, 80 % inside the central 10 % of the image"),
}


def survey_bytes(N, I, P, T, K):
    """SURVEY.md section 8(d) compulsory-traffic model (the coverage contract's per-unit figures):
    B_fwd = N(A+48+8+20) + I(12+24+8+44) + 24P + 16T,  B_bwd = 84 I + 28 P + (2A+40) N,
    B_adam = 28 (A/4) N, split per kernel of this build.  SSIM is not part of 8(d): layout bytes."""
    A = 44 + 12 * K
    return {"project_bin_sort": N * (A + 48 + 8 + 20) + I * (12 + 24 + 8) + 16 * T,
            "raster_fwd": 44 * I + 24 * P,
            "ssim": P * (24 + 36 + 36 + 12 + 24),
            "raster_bwd": 84 * I + 28 * P,
            "project_bwd": (2 * A + 40) * N,
            "adam": 28 * (A // 4) * N}


def algorithmic_bytes(N, I, P, T, K):
    """Compulsory-traffic model of THIS build's data layout, bytes/step."""
    A = 44 + 12 * K                      # parameter bytes per Gaussian
    b = {}
    # K1 (+ fused tile count): params in, record out; rank w/r, pair w/r, sorted id w, rect re-read by fill
    b["project_bin_sort"] = N * (A + 48) + N * 16 + I * (4 + 4 + 8 + 8 + 4) + 12 * T
    b["raster_fwd"] = I * (4 + 48) + P * (24 + 4) + 8 * T        # + the per-pixel stop position (round 4)
    b["ssim"] = P * (24 + 36 + 36 + 12 + 24)
    b["raster_bwd"] = I * (4 + 48 + 48) + P * (24 + 4 + 12 + 28) + 8 * T
    b["project_bwd"] = N * (A + 48 + A) + I * 48
    b["adam"] = 28 * (A // 4) * N
    return b


def measured_traffic(kernel_key, N, W, H, deg):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*pmc_step_cfg3.json: FETCH_SIZE / WRITE_SIZE collected in separate --pmc passes on this
    same workload and corrected as MI355X_MICROARCH.md prescribes).  None if no profile matches."""
    if (N, W, H, deg) != (1_000_000, 1920, 1080, 3):
        return None, None
    import glob
    names = {"raster_bwd": "k_raster_bwd", "raster_fwd": "k_raster_fwd<false>", "adam": "k_adam"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_step_cfg3.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            h = d["hbm_bytes_per_launch_corrected"]
            key = next(k for k in h if k.startswith(names[kernel_key].split("<")[0]) and "<true" not in k)
            return int(h[key]["total_bytes"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def calibration(dev):
    """~2 s: what THIS box does right now, so that numbers from two boxes can be told apart (the
    builder's 1019.7 vs the driver's 938.5 iters/s could not be attributed).  (1) `valu_fma_stream_ginst_s`: a plain
    v_fma_f32 stream at 4 waves per SIMD (tgs_calib_fma_stream) in 1e9 wave instructions / s -- the vector pipes under
    the box's power governor; K6 / K7 are VALU-bound and move with it.  (2) `copy_ceiling_GBs`: a device-to-device copy of
    1 GiB (read + write bytes / time) -- the HBM ceiling the fused optimizer kernel and the front half move with.
    (3) sclk / mclk from rocm-smi if it is readable."""
    import ctypes as C
    import shutil
    import subprocess
    from touch_gs_amd import _lib
    lib = _lib.load()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    sink = torch.zeros(1, device=dev)
    n = C.c_int64(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rates = []
    for it in range(6):     # the first two warm the clocks
        a, b = ev(), ev()
        a.record()
        _lib.check(lib.tgs_calib_fma_stream(20000, _lib.ptr(sink), C.byref(n), stream), "tgs_calib_fma_stream")
        b.record()
        torch.cuda.synchronize()
        rates.append(n.value / (a.elapsed_time(b) * 1e-3) / 1e9)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)     # 1 GiB
    dst = torch.empty_like(src)
    cps = []
    for it in range(6):
        a, b = ev(), ev()
        a.record()
        dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        cps.append(2 * src.numel() * 4 / (a.elapsed_time(b) * 1e-3) / 1e9)
    del src, dst
    out = {"valu_fma_stream_ginst_s": round(sorted(rates[2:])[2], 1), "valu_fma_stream_runs": [round(r, 1) for r in rates],
           "valu_fma_stream_peak_2cycle_ginst_s": round(1024 * 2.4 / 2, 1),
           "copy_ceiling_GBs": round(sorted(cps[2:])[2], 1), "copy_runs_GBs": [round(c, 1) for c in cps]}
    exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
    if exe:
        try:
            txt = subprocess.run([exe, "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
            card = next(iter(json.loads(txt).values()))
            out["rocm_smi_clocks"] = {k: v for k, v in card.items() if "sclk" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()}
        except Exception as ex:  # noqa: BLE001
            out["rocm_smi_clocks"] = {"error": repr(ex)[:200]}
    return out


def traffic_in_run(kernel_key, timeout_s=240):
    """HBM bytes per launch of the dominant kernel MEASURED NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE:
    they do not fit one pass) over three steps of this workload in a child process (tools/step_run.py), corrected as
    MI355X_MICROARCH.md's HBM section prescribes (gfx950: FETCH_SIZE x 2; both counters in KiB).  None if rocprofv3 is
    not on PATH, the tool is missing, or a pass fails / times out -- the caller then falls back to the committed profile."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    tool = os.path.join(ROOT, "tools", "step_run.py")
    if not exe or not os.path.exists(tool):
        return None
    names = {"raster_bwd": "k_raster_bwd(", "raster_fwd": "k_raster_fwd", "adam": "k_adam", "project_bwd": "k_project_bwd"}
    want = names.get(kernel_key)
    if want is None:
        return None
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tgs_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, tool, "3", "separate"],
                           cwd=ROOT, env=env, capture_output=True, timeout=timeout_s, check=True)
            acc = []
            for f in glob.glob(d + "/*/*counter_collection.csv"):
                for r in csv.DictReader(open(f)):
                    if want in r["Kernel_Name"] and "quad" not in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        acc.append(float(r["Counter_Value"]))
            if not acc:
                return None
            vals[ctr] = sum(acc) / len(acc)
        except Exception:  # noqa: BLE001
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return {"read_bytes": int(rd), "write_bytes": int(wr), "total_bytes": int(rd + wr)}


def valu_utilisation(kern_ms, N, W, H, deg):
    """Second roofline for the two compositing kernels, which are VALU- not HBM-bound: VALU
    wave-instructions per launch (SQ_INSTS_VALU from the committed PMC pass on this workload)
    divided by the measured time, against BOTH issue limits of 1024 SIMD32s @ 2.4 GHz: one wave64
    instruction per 2 cycles (MI355X_MICROARCH.md) and per 2.57 cycles (what a plain v_fma_f32 stream
    reaches in tools/ubench/pk_fma.hip: 122 TFLOP/s).  The instruction counts come from the committed
    profile named in `source`, not from this run."""
    if (N, W, H, deg) != (1_000_000, 1920, 1080, 3):
        return None
    import glob
    peak, peak_spec = 1024 * 2.4e9 / 2.57, 1024 * 2.4e9 / 2.0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_step_cfg3.json")), reverse=True):
        try:
            with open(path) as f:
                c = json.load(f)["counters"]
            out = {"peak_wave_instr_per_s": round(peak, -9), "peak_spec_2cycle_wave_instr_per_s": round(peak_spec, -9),
                   "source": os.path.relpath(path, ROOT)}
            for key, name in (("raster_fwd", "k_raster_fwd"), ("raster_bwd", "k_raster_bwd")):
                n = c[next(k for k in c if k.startswith(name) and "<true" not in k)]["SQ_INSTS_VALU"]
                rate = n / (kern_ms[key] * 1e-3)
                out[key] = {"valu_wave_instr": int(n), "frac_of_issue_peak": round(rate / peak, 3),
                            "frac_of_spec_2cycle_peak": round(rate / peak_spec, 3)}
            return out
        except Exception:
            continue
    return None


def cpu_baseline(N, W, H, deg, seed, frac=None, clustered=False):
    """Build's own scalar C restatement (oracle/ref_raster.c, fp32, OpenMP over tiles) timed on the
    host cores: full projection / binning / projection-backward, compositing fwd+bwd on the first
    1/frac of the tiles and extrapolated.  The reference's rasterizer source is unavailable."""
    from oracle.ref_c import RefC
    from oracle import torch_oracle as O
    import numpy as np
    R = RefC("f32")
    if frac is None:  # ~10-30 s of CPU work: whole image on a many-core host, a band of tiles otherwise
        frac = max(1, 64 // max(R.num_threads(), 1))
    if clustered:   # the oracle's generator restates the iid recipe only: take the product's scene tensors
        from touch_gs_amd.scene import synthetic_gaussians
        P, c = synthetic_gaussians(N, W, H, deg, seed, clustered=True)
    else:
        P, c = O.synthetic_scene(N, W, H, deg, seed, dtype=torch.float32)
    n = lambda t: t.numpy()
    cam = R.cam_block(O.orbit_viewmat(0, 8).numpy(), c["fx"], c["fy"], c["cx"], c["cy"])
    t0 = time.perf_counter()
    pr = R.project_fwd(n(P["means"]), n(P["log_scales"]), n(P["quats"]), n(P["opac_logit"]), n(P["sh"]), deg, cam, W, H)
    t1 = time.perf_counter()
    gid, ts = R.bin_sort(pr["rect"], pr["tiles_hit"], pr["depth"], W, H)
    t2 = time.perf_counter()
    T = (len(ts) - 1)
    tr = (0, max(T // frac, 1))
    f = R.blend_fwd(pr["xy"], pr["conic"], pr["opac"], pr["rgb"], pr["depth"], gid, ts, cam, W, H, tile_range=tr)
    t3 = time.perf_counter()
    ones3 = np.ones((H, W, 3), np.float32)
    ones1 = np.ones((H, W), np.float32)
    b = R.blend_bwd(pr["xy"], pr["conic"], pr["opac"], pr["rgb"], pr["depth"], gid, ts, cam, W, H,
                    f["final_T"], f["final_idx"], ones3, ones1, ones1, tile_range=tr)
    t4 = time.perf_counter()
    R.project_bwd(n(P["means"]), n(P["log_scales"]), n(P["quats"]), n(P["opac_logit"]), n(P["sh"]), deg, cam, W, H,
                  pr["radius"], b["v_xy"], b["v_conic"], b["v_opac"], b["v_rgb"], b["v_depth"])
    t5 = time.perf_counter()
    scale = T / (tr[1] - tr[0])
    est = (t1 - t0) + (t2 - t1) + (t3 - t2) * scale + (t4 - t3) * scale + (t5 - t4)
    return dict(value=1.0 / est, unit="train iters/s (fwd+bwd, no SSIM/Adam)", cores=R.num_threads(), kind="port",
                sample=(f"build's own C restatement of the published algorithm (oracle/ref_raster.c, fp32, OpenMP; "
                        f"reference rasterizer source unavailable): full project/bin-sort/project-bwd + compositing "
                        f"fwd+bwd on tiles [0,{tr[1]}) of {T}, extrapolated x{scale:.1f}; measured "
                        f"{(t5 - t0):.1f} s CPU wall"),
                seconds=dict(project=t1 - t0, bin_sort=t2 - t1, blend_fwd_sample=t3 - t2, blend_bwd_sample=t4 - t3,
                             project_bwd=t5 - t4))


def cpu_baseline_torch(cfg_name="cfg2", full=False):
    """north_star: "the reference's pure-PyTorch CPU rasterizer timed on the same box's host cores".  The reference's
    rasterizer source is absent (empty submodule), so this is the build's vectorised PyTorch restatement of the published
    algorithm (oracle/torch_oracle.py, fp32 here; its autograd is the backward) -- forward + backward of configs[1], one view,
    on torch's intra-op thread pool.  ``full``: the whole frame (55 + 107 s on 128 cores:
    profiles/r5_a_bench_cfg3.json).  Default: a BOUNDED sample -- all 100 k Gaussians through the centre quarter of the frame
    (400 x 400 of 800 x 800: a quarter of the tiles, ~40 s) with the whole-frame rate extrapolated x 1/4 and labelled so --
    so that the default `python bench.py` stays within a few minutes."""
    from oracle import torch_oracle as O
    c = CONFIGS[cfg_name]
    N, W, H, deg = c["gaussians"], c["width"], c["height"], 3
    P, intr = O.synthetic_scene(N, W, H, deg, c["seed"], dtype=torch.float32)
    frac = 1.0
    if not full:       # centre window: same intrinsics, principal point shifted, a quarter of the pixels
        intr = dict(intr, W=W // 2, H=H // 2, cx=intr["cx"] - W // 4, cy=intr["cy"] - H // 4)
        frac = 0.25
    cam = O.Camera(viewmat=O.orbit_viewmat(0, 8, dtype=torch.float32), **intr)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    t0 = time.perf_counter()
    out, _, _, _ = O.render(Pg["means"], Pg["log_scales"], Pg["quats"], Pg["opac_logit"], Pg["sh"], cam, deg)
    t1 = time.perf_counter()
    (out["rgb"].sum() + out["depth_acc"].sum()).backward()
    t2 = time.perf_counter()
    what = "in full" if full else ("the centre quarter of the frame (all Gaussians projected, a quarter of the tiles composited); "
                                   "value = whole-frame rate EXTRAPOLATED as sample rate x 1/4 (measured in full on this "
                                   "class of box: 0.0062 iters/s, profiles/r5_a_bench_cfg3.json; --cpu-baseline-torch-full)")
    return dict(value=round(frac / (t2 - t0), 5), unit="train iters/s (fwd+bwd, no SSIM/Adam)", cores=torch.get_num_threads(),
                kind="port", config=c["label"], extrapolated=not full,
                sample=f"build's PyTorch-CPU oracle (oracle/torch_oracle.py, fp32, autograd backward), {c['label']}, one view, "
                       f"{what}: forward {t1 - t0:.1f} s + backward {t2 - t1:.1f} s",
                seconds=dict(forward=round(t1 - t0, 2), backward=round(t2 - t1, 2)))


def train_quality_run(full: bool = False):
    """Does the step the headline times TRAIN?  The whole scripts/train_bunny_real.sh sequence on the known-geometry
    capture (touch_gs_amd/analytic_scene.py): raw capture -> prepare -> touch_gs_amd.train -> run_eval, held-out views
    only.  Default: reduced size (24 views at 640 x 360, 4000 iterations per run; what tests/test_gpu_train_quality.py
    asserts on).  ``full``: 100 views at 1280 x 720, 30 000 iterations, the reference's two flag sets with and without the
    depth term (tools/train_quality.py; ~6 min; committed results: profiles/r5_train_quality.json)."""
    import tempfile
    from touch_gs_amd import analytic_scene as A
    root = tempfile.mkdtemp(prefix="tq_")
    keys = ("psnr", "ssim", "depth_mse", "gt_depth_mse", "gt_object_depth_mse", "gt_depth_mse_true_object_mask", "exact_depth_mse", "exact_object_depth_mse",
            "exact_object_depth_median_abs_m", "gaussian_count", "iters", "split", "train_wall_s", "iters_per_s_wall")
    if full:
        t0 = time.perf_counter()
        cap = A.write_raw_capture(root, n_views=100, device="cuda")
        A.prepare_capture(root, 0.08)
        runs = {f"{f}:{d}": A.train_and_eval(root, f, d == 1, iters=30000) for f in ("bunny_real", "block") for d in (1, 0)}
        r = dict(capture=cap, runs=runs, total_s=round(time.perf_counter() - t0, 1))
    else:
        r = A.quick_quality(root)
    import shutil
    shutil.rmtree(root, ignore_errors=True)
    out = {"size": "100 views 1280x720, 30000 iterations" if full else "24 views 640x360, 4000 iterations (reduced)",
           "total_s": r["total_s"], "gpis_rmse_m": r["capture"]["gpis_rmse_m"], "held_out_views_only": True,
           "runs": {k: {m: (round(v[m], 5) if isinstance(v[m], float) else v[m]) for m in keys if m in v}
                    for k, v in r["runs"].items()},
           "key": "flag set (scripts/train_block_data.sh:50 at 0.8 split | scripts/train_bunny_real.sh:52 few views) : "
                  "1 = with the depth term, 0 = RGB only; depth errors in the dataparser's scaled frame"}
    few, rgb = r["runs"].get("bunny_real:1"), r["runs"].get("bunny_real:0")
    if few and rgb:
        out["few_view_depth_mse_with_over_without"] = round(few["depth_mse"] / rgb["depth_mse"], 3)
    return out


def densify_run(N, W, H, deg, seed, dev, views, steps=600, clustered=False):
    """The workload the reference's method actually runs for the first half of its 30 000 iterations
    (a densifying Splatfacto), next to the steady state the headline value is quoted on:
    the same scene and views, refinement every 100 steps from step 100 on (clone / split / cull on the
    Splatfacto thresholds, opacity reset every 3rd refinement so that the window holds some), SH
    degree ramp 0 -> `deg` (one band per 100 steps), sync-free intersection budget (refinements are its
    barrier points), colour prefetch, children placed behind their parents + a full Morton re-sort every
    4th refinement.  Everything is inside the timed window, refinements and re-sorts included."""
    from touch_gs_amd.densify import DensifyConfig
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import synthetic_gaussians
    P, _ = synthetic_gaussians(N, W, H, deg, seed, clustered=clustered)
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    cfg = ModelConfig(sh_degree=deg, sh_degree_interval=100, depth_loss_mult=0.2,
                      depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS", uncertainty_weight=1.0, spatial_sort=True,
                      resort_every_refines=4)
    m = DepthGaussianSplattingModel(cfg, params)
    m.spatial_sort()
    m.enable_densification(DensifyConfig(warmup_length=100, refine_every=100, reset_alpha_every=3,
                                         max_gaussians=int(1.6 * N)))
    m.enable_speculative_budget()
    n0 = m.params.N
    W0 = 110   # warm-up: through the first refinement (step 100), whose torch ops pay ~0.1 s of one-off first-use
    for i in range(W0):   # initialisation per process (tools/refine_cost.py: 111 ms against ~3 ms for every later one)
        m.train_step(views[i % len(views)], next_view=views[(i + 1) % len(views)])
    m.flush()
    # the re-sort's own one-off: optim.balanced_order (long-run Gaussians dealt over the groups, round 6) touches a
    # dozen torch ops for the first time in the process (~0.3 s of lazy kernel loading, measured); a 30 000-step run
    # never notices, a 300-step window would be half that.  Every later re-sort (3 - 6 ms at 1 M) is inside the window.
    m.flush()
    m.spatial_sort()
    torch.cuda.synchronize()
    refines, t0 = [], time.perf_counter()
    for i in range(W0, W0 + steps):
        m.train_step(views[i % len(views)], next_view=views[(i + 1) % len(views)])
        if getattr(m, "last_refine", None) is not None and (not refines or refines[-1] is not m.last_refine):
            refines.append(m.last_refine)
    m.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 2), "unit": "iters/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "gaussians_start": n0, "gaussians_end": m.params.N, "refinements": len(refines),
            "cloned": sum(r["cloned"] for r in refines), "split": sum(r["split"] for r in refines),
            "culled": sum(r["culled"] for r in refines), "opacity_resets": sum(int(r["opacity_reset"]) for r in refines),
            "replayed_steps": getattr(m, "speculative_replays", 0), "final_sh_degree": m.active_sh_degree(),
            "workload": "same scene and views; refine every 100 steps from step 100, SH ramp 1 band / 100 steps, "
                        "sync-free budget, colour prefetch; timed from step 110 (after the first refinement and one "
                        "re-sort, which pay the process's one-off torch first-use costs), later refinements and re-sorts "
                        "inside the timed window"}


def touch_scene_run(dev, steps=3000, target=300_000, seeds=5000, W=1280, H=720, deg=3, n_views=30, seed=77, window=500):
    """The regime the reference trains in (`train_densify` above starts from 1 M Gaussians and
    SHRINKS): 1280 x 720 (reference utils/fuse_touch_vision.py:278), 30 orbit views of an object-centric target scene,
    a model seeded with `seeds` touch points that GROWS under Splatfacto's schedule with the trainer's defaults --
    DensifyConfig() (warm-up 500, refinement every 100 steps, opacity reset every 30 refinements), SH ramp, resolution
    schedule 2 / 250, sync-free budget, prefetches, Morton order, flags of scripts/train_block_data.sh:50.  Reported per
    `window` steps: iters/s, N, host time of the refinement steps, replayed steps.  tools/touch_scene_run.py adds the
    eager-vs-hipGraph comparison at small N and the same run through the trainer on disk."""
    from touch_gs_amd import train
    from touch_gs_amd.densify import DensifyConfig
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.scene import make_views
    t0 = time.perf_counter()
    views, D = make_views(target, W, H, deg, seed, dev, n_views, clustered=True)
    g = torch.Generator().manual_seed(1)
    pick = torch.randperm(target, generator=g)[:seeds].to(dev)
    pts = D["means"][pick].float().cpu()
    cols = ((D["sh"][pick, 0].float() * 0.28209479177387814 + 0.5).clamp(0, 1) * 255).cpu()
    params = train.init_params(seeds, (deg + 1) ** 2, dev, (pts, cols), seed=0, seed_fraction=1.0)
    cfg = ModelConfig(sh_degree=deg, depth_loss_mult=0.2, depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS",
                      uncertainty_weight=1.0, spatial_sort=True, num_downscales=2, resolution_schedule=250)
    m = DepthGaussianSplattingModel(cfg, params)
    m.spatial_sort()
    m.enable_densification(DensifyConfig())
    m.enable_speculative_budget()
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    windows, t_ref, n_ref = [], 0.0, 0
    tw = t_all = time.perf_counter()
    for step in range(steps):
        due = m.density.due(m.step + 1)
        if due:
            torch.cuda.synchronize(); tr = time.perf_counter()
        m.train_step(views[step % n_views], next_view=views[(step + 1) % n_views])
        if due:
            torch.cuda.synchronize(); t_ref += time.perf_counter() - tr; n_ref += 1
        if (step + 1) % window == 0:
            m.flush(); torch.cuda.synchronize()
            now = time.perf_counter()
            windows.append({"steps": f"{step + 2 - window}-{step + 1}", "iters_per_s": round(window / (now - tw), 1),
                            "N": m.params.N, "sh_degree": m.active_sh_degree(), "refinements": n_ref,
                            "refinement_steps_s": round(t_ref, 3), "replayed_steps": getattr(m, "speculative_replays", 0)})
            tw, t_ref, n_ref = now, 0.0, 0
    m.flush(); torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    ev = m.get_outputs(views[0].cam)
    return {"value": round(steps / total, 1), "unit": "iters/s", "steps": steps, "seconds": round(total, 2),
            "gaussians_start": seeds, "gaussians_end": m.params.N,
            "replayed_steps": getattr(m, "speculative_replays", 0), "setup_s": round(setup_s, 1),
            "psnr_view0": round(float(-10 * torch.log10(((ev["rgb"] - views[0].rgb) ** 2).mean())), 2),
            "workload": f"{W}x{H}, SH {deg} with ramp, {n_views} orbit views of a {target}-Gaussian object-centric target, start "
                        f"from {seeds} touch seeds, DensifyConfig defaults, resolution schedule 2 / 250, sync-free budget, "
                        "prefetches; the first window includes the process's one-off first-refinement cost (~0.3 s)",
            "windows": windows}, views, (pts, cols)


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks through
    torch.distributed.run on 127.0.0.1 (a free port), pass their chatter to stderr and print exactly
    the one JSON line rank 0 produced.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = r.stdout.splitlines()
    js = [l for l in lines if l.startswith("{") and l.rstrip().endswith("}")]
    for l in lines:
        if not js or l is not js[-1]:
            print(l, file=sys.stderr)
    if js:
        _emit(json.loads(js[-1]))
    return r.returncode if (r.returncode != 0 or js) else 1


_REAL_STDOUT = None


def _emit(line: dict) -> None:
    """The ONE JSON line of the run, on the real stdout (see main)."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)   # RCCL's banner (C stdio) first, the JSON line last
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(line) + "\n").encode())


def main():
    """Contract: rank 0 prints ONE JSON line on stdout.  Everything else the run prints on the way -- the trainer's
    progress lines and eval dictionaries of the train_quality runs (JSON-shaped themselves), RCCL's banner, warnings of
    child processes -- is sent to stderr by pointing file descriptor 1 at stderr for the duration of the run; _emit
    writes the result to the saved descriptor."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    try:
        _main()
    finally:
        sys.stdout.flush()
        os.dup2(_REAL_STDOUT, 1)
        os.close(_REAL_STDOUT)
        _REAL_STDOUT = None


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cpu-baseline-torch-full", action="store_true",
                    help="time the PyTorch-CPU oracle on the whole configs[1] frame (~3 min on 128 cores) instead of its centre quarter")
    ap.add_argument("--no-list-hint", action="store_true",
                    help="launch every sort class every frame (A/B of tgs_bin_sort's max_list_hint)")
    ap.add_argument("--repeats", type=int, default=4, help="further runs of the timed loop after the headline one (value_repeats)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the ~2 s box calibration block")
    ap.add_argument("--no-traffic-run", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes in a child process, ~1 min); "
                         "the committed profile is used instead")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3",
                    help="workload: cfg3 = the configuration BASELINE.json's metric is quoted on (default)")
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-busbw-sweep", action="store_true", help="skip the message-size sweep of the exchange (N > 1)")
    ap.add_argument("--dp-report", action="store_true",
                    help="N > 1: after the headline loop (TGS_DP_TRANSPORT, default rccl) time the same loop over every "
                         "transport back to back -- RCCL, peer stores with flags behind the kernel boundary, peer stores "
                         "with in-kernel flags -- each followed by the replica check: dp_exchange.transports")
    ap.add_argument("--loop-only", action="store_true",
                    help="run ONLY warm-up + the timed loop (no other layout, no per-kernel repetitions, no render-only / "
                         "densifying / CPU measurements) and print a short JSON line: the command tools/collect_round.sh "
                         "puts under rocprofv3, so that the kernel statistics are those of the timed loop and "
                         "sum(average x calls per step) reproduces ms_per_step (profiles/r4_*_reconcile.json)")
    ap.add_argument("--no-densify-run", action="store_true",
                    help="skip the densifying-training measurement reported as `train_densify` (N = 1 only)")
    ap.add_argument("--train-quality", choices=("reduced", "full", "off"), default="reduced",
                    help="end-to-end training quality on the known-geometry capture, reported as `train_quality` (N = 1): "
                         "reduced = 24 views at 640x360, ~30 s; full = 100 views at 720p, 30 000 iterations x 4 runs, ~6 min")
    ap.add_argument("--touch-scene-run", action="store_true",
                    help="also run rounds 3-4's growing-scene throughput measurement (`train_touch_scene`: machinery, not quality)")
    ap.add_argument("--ssim-pipeline", action="store_true",
                    help="SSIM on a second stream pipelined by image bands behind K7 (measured slower; off by default)")
    ap.add_argument("--no-color-prefetch", action="store_true",
                    help="K1 always evaluates the SH rows itself (off: the previous step's optimizer kernel does)")
    ap.add_argument("--layout", choices=("morton", "asis"), default="morton",
                    help="memory order of the Gaussians: morton = model.spatial_sort() at start-up (the "
                         "framework's default layout), asis = the order the scene generator emits (random)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    preset = CONFIGS[args.config]
    custom = []
    for k in ("gaussians", "width", "height", "seed"):
        if getattr(args, k) is None:
            setattr(args, k, preset[k])
        elif getattr(args, k) != preset[k]:
            custom.append(f"{k}={getattr(args, k)}")
    clustered = bool(preset.get("clustered"))
    workload = preset["label"] if not custom else f"custom ({', '.join(custom)}; base {args.config})"
    workload += (", full Touch-GS train loop (RGB L1/SSIM + tactile depth/uncertainty loss), "
                 "one view per rank per iter")

    from touch_gs_amd import ops, parallel
    from touch_gs_amd.model import DepthGaussianSplattingModel, ModelConfig
    from touch_gs_amd.optim import GaussianParams
    from touch_gs_amd.scene import make_view, synthetic_gaussians

    dp = parallel.init_from_env()
    if dp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={dp.world}; launch with torch.distributed.run")
    dev = torch.device("cuda", dp.local_rank)
    torch.cuda.set_device(dev)
    N, W, H, deg = args.gaussians, args.width, args.height, args.sh_degree
    K = (deg + 1) ** 2

    # ---- setup (untimed): scene, replicas, views, intersection capacity ----
    P, intr = synthetic_gaussians(N, W, H, deg, args.seed, clustered=clustered)
    params = GaussianParams.from_tensors(*[P[k].to(dev) for k in GaussianParams.NAMES])
    cfg = ModelConfig(sh_degree=deg, sh_degree_interval=0, depth_loss_mult=0.2,
                      depth_loss_type="DEPTH_UNCERTAINTY_WEIGHTED_LOSS", uncertainty_weight=1.0,
                      pipeline_ssim=args.ssim_pipeline)
    model = DepthGaussianSplattingModel(cfg, params)
    views = [make_view(N, W, H, deg, args.seed, dev, view=v, n_views=args.views, clustered=clustered)
             for v in range(args.views)]
    for v in views:
        v.valid_count()
    n_isect, n_need, n_longest = [], [], []
    for v in views:  # size the intersection buffers once; no host sync inside the timed region
        sp = ops.project_fwd(v.cam, params.means, params.log_scales, params.quats, params.opac_logit, params.sh, deg)
        b = ops.IntersectBudget()
        ops.bin_sort(v.cam, sp, b)
        n_isect.append(b.last_n)
        n_need.append(b.last_need)   # capacity under the per-XCD split of the pair index space (tgs.h)
        n_longest.append(b.last_longest)
    # ... and bound the longest tile list the same way (1.25 x the longest list of any view + 32: the capacity's margin; the
    # trainer's speculative budget learns its bound from the frames it has settled): the sort launches for longer list classes
    # are not issued; a frame that broke the bound would be void and budget.check() after the timed region would raise
    model.budget = ops.IntersectBudget(capacity=int(max(n_need) * 1.25) + 4096, sync=False,
                                       max_list_hint=-1 if args.no_list_hint else int(1.25 * max(n_longest)) + 32)
    del sp
    torch.cuda.empty_cache()

    def step(i):
        # single process: the view of the following step is known, so the optimizer kernel of this step
        # also evaluates the colours the updated Gaussians show to it (colour prefetch; same results)
        # (data parallel: this rank's next view -- the geometry Adam of the step then runs its K1, front prefetch)
        nxt = None if args.no_color_prefetch else views[dp.views_for_step(i + 1, len(views))]
        model.train_step(views[dp.views_for_step(i, len(views))], dp if dp.active else None, next_view=nxt)

    done = [0]

    def timed(n_warm, n_steps):
        """n_warm untimed + n_steps timed steps, barrier + synchronize on both sides, max over ranks."""
        for _ in range(n_warm):
            step(done[0]); done[0] += 1
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(done[0]); done[0] += 1
        torch.cuda.synchronize()
        dp.barrier()
        torch.cuda.synchronize()
        return dp.max_over_ranks(time.perf_counter() - t0)

    value_asis = None
    if args.layout == "morton" and args.loop_only:
        model.spatial_sort()
    elif args.layout == "morton":
        # The framework keeps its Gaussians in 3-D Morton order (model.spatial_sort() at start-up and
        # after densification): a pure re-layout -- the train step computes the same thing on permuted
        # rows -- that lets the binning count per (group, tile).  The same measurement on the order the
        # scene generator emits is reported next to it as `value_asis_layout`.
        k_asis = min(args.steps, 50)
        value_asis = k_asis * dp.world / timed(min(args.warmup, 20), k_asis)
        model.spatial_sort()
    elapsed = timed(args.warmup, args.steps)
    # the same timed loop again, back to back (no further warm-up): `value` stays the FIRST loop -- what the driver's
    # contract times -- and `value_repeats` / `value_spread` say how far a 20-step loop after 5 warm-ups is from the
    # box's steady state (clock ramp, first-touch effects): VERDICT r5 next #6
    rep_elapsed = [elapsed] + [timed(0, args.steps) for _ in range(max(args.repeats, 0))]
    model.budget.check()  # raises if ANY timed frame overflowed its intersection buffer (sticky word)
    replicas_identical = None
    if dp.world > 1:
        dp.assert_replicas_identical(model.params.flat)   # raises on divergence
        replicas_identical = True
    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps * dp.world / elapsed
    if args.loop_only:
        if dp.rank == 0:
            _emit({"metric": "train iters/s (timed loop only)", "value": round(value, 3), "unit": "iters/s",
                   "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(ms_per_step, 4), "launched_steps": done[0],
                   "config": {"workload": workload, "name": args.config, "layout": args.layout}})
        dp.barrier()
        return

    # ---- collective times of the exchange (outside the timed region; every rank takes part) ----
    comm = None
    if dp.world > 1:
        dp.timing = True
        for i in range(3):
            step(done[0]); done[0] += 1
        comm = dp.comm_report()
        dp.timing = False
    # --dp-report: the SAME timed loop over every transport, back to back on this job's ranks (every rank takes part; all
    # ranks walk the list in the same order).  A transport that fails (a peer wait that times out poisons its steps and
    # raises at the check) is reported with its error and the next one starts from a fresh PeerExchange.
    transports = None
    if dp.world > 1 and args.dp_report:
        transports = {}
        first = (dp.transport, os.environ.get("TGS_PEER_SAFE_FLAGS", "1"))
        for name, tr, safe in (("rccl", "rccl", None), ("ipc_flags_behind_kernel_boundary", "ipc", "1"),
                               ("ipc_flags_in_kernel", "ipc", "0")):
            if dp.peer is not None:
                dp.check_transport(); dp.peer.close(); dp.peer = None
            dp.transport = tr
            if safe is not None:
                os.environ["TGS_PEER_SAFE_FLAGS"] = safe
            try:
                k = min(args.steps, 100)
                el = timed(min(args.warmup, 20), k)
                dp.check_transport()
                model.budget.check()
                dp.assert_replicas_identical(model.params.flat)
                transports[name] = {"ms_per_step": round(el / k * 1e3, 4), "iters_per_s": round(k * dp.world / el, 2),
                                    "replicas_identical": True, "steps": k}
                if dp.peer is not None:
                    transports[name]["memory_kind"] = dp.peer.memory_kind
            except Exception as ex:  # noqa: BLE001 -- the report must survive one transport failing
                transports[name] = {"error": repr(ex)[:400]}
        if dp.peer is not None:
            try:
                dp.check_transport(); dp.peer.close()
            except Exception:  # noqa: BLE001
                pass
            dp.peer = None
        dp.transport = first[0]
        os.environ["TGS_PEER_SAFE_FLAGS"] = first[1]
    # message-size sweep of the exchange primitives on this job's ranks (every rank takes part): replaces the link
    # figures assumed in DESIGN.md section 6 by measurements the moment a multi-GPU node runs this
    busbw = None
    if dp.world > 1 and not args.no_busbw_sweep:
        dp.check_transport()
        busbw = dp.busbw_sweep(dev)

    # ---- per-kernel timing of the same step (HIP events on the launch stream), rank 0 ----
    out = None
    if dp.rank == 0:
        ev = lambda: torch.cuda.Event(enable_timing=True)
        names = ["project_bin_sort", "raster_fwd", "ssim", "raster_bwd", "project_bwd", "adam"]
        reps = min(args.steps, 20)
        p = params
        # the same measurement on the other views of the timed loop (5 repetitions each): the loop cycles through all
        # of them and view 0 is the fullest, so `kernel_ms` (view 0) overstates the loop's K6 / K7; `kernel_ms_view_mean`
        # and `roofline.frac_view_mean` (mean pair count) are what the timed loop averages
        per_view = []
        for vi in range(len(views) - 1, -1, -1):        # view 0 last: its tensors feed the measurements below
          view = views[vi]
          I = n_isect[vi]
          acc = {k: [] for k in names}
          all_ev = []
          for _ in range(reps if vi == 0 else 5):     # enqueued back to back; ONE synchronisation after the last repetition, so
              e = [ev() for _ in range(7)]   # that no kernel starts on an idle GPU behind a host round trip
              e[0].record()
              sp, _, gb, ts, sg, _ = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit,
                                                          p.sh, deg, model.budget)
              e[1].record()
              rgb, dacc, fT, fidx = ops.rasterize_fwd(view.cam, sp, sg, ts)
              e[2].record()
              ssim_sum, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-cfg.ssim_lambda / (3 * H * W))
              e[3].record()
              partials, tl = ops.rasterize_bwd(view.cam, sp, gb, sg, ts, rgb, dacc, fT, v_rgb=v_img,
                                               loss=model.loss_spec(view), want_tile_loss=True)
              e[4].record()
              ops.project_bwd(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, sp, gb, partials,
                              out=p.grad_views())
              e[5].record()
              p.grad.zero_()  # keep the scene fixed while profiling
              e[6].record()
              model.optimizer.step()
              e7b = ev(); e7b.record()
              all_ev.append((e, e7b))
          torch.cuda.synchronize()
          for e, e7b in all_ev:
              for j, k in enumerate(names[:5]):
                  acc[k].append(e[j].elapsed_time(e[j + 1]))
              acc["adam"].append(e[6].elapsed_time(e7b))
          # median over the repetitions: a single disturbed launch must not move the roofline line
          kern_ms = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
          per_view.append(kern_ms)
        kern_ms_mean = {k: sum(d[k] for d in per_view) / len(per_view) for k in names}
        I_mean = sum(n_isect) / len(n_isect)
        # the front half as the single-process step runs it: K1 on colours prefetched by the previous
        # step's fused optimizer kernel (one real fused step arms them; reported next to the plain form)
        front_pre_ms = front_fin_ms = None
        if not dp.active and not args.no_color_prefetch and model.optimizer.can_fuse_with_backward(deg):
            pf = ops.ColorPrefetch(N, dev).arm(view.cam, deg)
            model.optimizer.backward_and_step(view.cam, deg, sp, gb, partials, prefetch=pf)
            evp = [(ev(), ev()) for _ in range(reps)]
            for a, b in evp:
                a.record()
                ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, model.budget,
                                     colors=pf)
                b.record()
            torch.cuda.synchronize()
            front_pre_ms = sorted(a.elapsed_time(b) for a, b in evp)[reps // 2]
            # ... and with the front prefetch on top (the optimizer kernel has also run this view's K1): scan, fill, sort
            if getattr(model, "front_prefetch", False):
                evf = [(ev(), ev()) for _ in range(5)]
                for a, b in evf:
                    pf = ops.ColorPrefetch(N, dev).arm(view.cam, deg, ops.FrontBuffers(view.cam, N, model.budget.initial(N),
                                                                                       False, dev), model.budget)
                    model.optimizer.backward_and_step(view.cam, deg, sp, gb, partials, prefetch=pf)
                    a.record()
                    ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, model.budget,
                                         colors=pf)
                    b.record()
                torch.cuda.synchronize()
                front_fin_ms = sorted(a.elapsed_time(b) for a, b in evf)[2]
        # render-only throughput (K1..K6)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        rr = 20
        for _ in range(rr):
            sp, _, gb, ts, sg, _ = ops.project_bin_sort(view.cam, p.means, p.log_scales, p.quats, p.opac_logit,
                                                        p.sh, deg, model.budget)
            ops.rasterize_fwd(view.cam, sp, sg, ts, want_stop=False)   # render only: no backward, no stop positions
        torch.cuda.synchronize()
        render_ms = (time.perf_counter() - r0) / rr * 1e3
        # ... and cycling through all the views of the timed loop (view 0 above is the fullest one)
        r0 = time.perf_counter()
        for i in range(3 * len(views)):
            v = views[i % len(views)]
            spv, _, _, tsv, sgv, _ = ops.project_bin_sort(v.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, model.budget)
            ops.rasterize_fwd(v.cam, spv, sgv, tsv, want_stop=False)
        torch.cuda.synchronize()
        render_mean_ms = (time.perf_counter() - r0) / (3 * len(views)) * 1e3
        del spv, tsv, sgv

        T = view.cam.num_tiles
        longest = int((ts[1:] - ts[:-1]).max())
        ab = algorithmic_bytes(N, I, W * H, T, K)   # this build's data layout
        sb = survey_bytes(N, I, W * H, T, K)        # the contract's per-unit bytes
        dom = max(kern_ms, key=kern_ms.get)
        achieved = sb[dom] / (kern_ms[dom] * 1e-3) / 1e9
        achieved_layout = ab[dom] / (kern_ms[dom] * 1e-3) / 1e9
        # bytes of the kernels the TIMED LOOP runs: single process = K8 and Adam fused (the gradient never reaches HBM:
        # no A N write by K8, no A N read by Adam); data parallel = the unfused layout
        A_bytes = 44 + 12 * K
        fused_loop = not dp.active and model.optimizer.can_fuse_with_backward(deg)
        step_bytes = sum(ab.values()) - (2 * A_bytes * N if fused_loop else 0)
        survey_step_bytes = sum(v for k_, v in sb.items() if k_ != "ssim")   # B_fwd + B_bwd + B_adam
        traffic, traffic_src = measured_traffic(dom, N, W, H, deg) if args.config == "cfg3" and not custom else (None, None)
        traffic_detail = None
        if args.config == "cfg3" and not custom and not args.no_traffic_run:
            torch.cuda.empty_cache()
            traffic_detail = traffic_in_run(dom)
            if traffic_detail is not None:
                traffic, traffic_src = traffic_detail["total_bytes"], "measured in this run (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, tools/step_run.py)"
        fwd_bwd_ms = sum(kern_ms[k] for k in names[:5] if k != "ssim")
        fwd_bwd_bytes = sum(sb[k] for k in names[:5] if k != "ssim")
        out = {
            "metric": "train iters/s (whole-job views/s) + render Mpix/s, 1M Gaussians @ 1080p"
                      if args.config == "cfg3" and not custom else "train iters/s (whole-job views/s) + render Mpix/s",
            "value": round(value, 3), "unit": "iters/s",
            "n_gpus": dp.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_repeats": [round(args.steps * dp.world / e, 2) for e in rep_elapsed],
            "value_spread": round((max(rep_elapsed) - min(rep_elapsed)) / sorted(rep_elapsed)[len(rep_elapsed) // 2], 4),
            "render_mpix_s": round(W * H / (render_ms * 1e-3) / 1e6, 1),
            "render_mpix_s_view_mean": round(W * H / (render_mean_ms * 1e-3) / 1e6, 1),
            "value_asis_layout": None if value_asis is None else round(value_asis, 3),
            "config": {"workload": workload, "name": args.config if not custom else "custom",
                       "gaussians": N, "width": W, "height": H, "sh_degree": deg, "views": args.views,
                       "intersections": I, "tiles": T, "longest_tile_list": longest,
                       "longest_tile_list_any_view": max(n_longest), "max_list_hint": model.budget.max_list_hint,
                       "parallelism": f"dp{dp.world}",
                       "layout": "morton (model.spatial_sort(): same scene, rows permuted)" if args.layout == "morton"
                                 else "as generated (random order)",
                       "depth_loss_type": cfg.depth_loss_type},
            "kernel_ms": {k: round(v, 4) for k, v in kern_ms.items()},
            "kernel_ms_view_mean": {k: round(v, 4) for k, v in kern_ms_mean.items()},
            "intersections_view_mean": int(I_mean),
            "project_bin_sort_prefetched_ms": None if front_pre_ms is None else round(front_pre_ms, 4),
            "front_half_after_front_prefetch_ms": None if front_fin_ms is None else round(front_fin_ms, 4),
            # dominant kernel: `achieved`/`frac` use SURVEY 8(d)'s algorithmic bytes (the contract);
            # the *_layout figures use this build's own record sizes
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch: measured in this run when rocprofv3 is available (two --pmc passes in a
                         # child process), else from the committed PMC passes of this workload; `traffic_source` says which
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_read_write": traffic_detail,
                         "algorithmic_bytes": sb[dom], "bytes_model": "SURVEY 8(d)",
                         # the same kernel averaged over the views the timed loop cycles through (mean pair count)
                         "frac_view_mean": round(survey_bytes(N, I_mean, W * H, T, K)[dom] / (kern_ms_mean[dom] * 1e-3) / 1e9
                                                 / HBM_PEAK_GBS, 4),
                         "achieved_layout": round(achieved_layout, 1), "frac_layout": round(achieved_layout / HBM_PEAK_GBS, 4),
                         "layout_bytes": ab[dom]},
            "step_roofline": {"algorithmic_bytes": step_bytes,
                              "bytes_model": "this build's layout, kernels of the timed loop"
                                             + (" (K8 + Adam fused: no gradient round trip)" if fused_loop else ""),
                              "survey_bytes_fwd_bwd_adam": survey_step_bytes,
                              "survey_frac_of_hbm_peak": round(survey_step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "achieved_GBs": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              "fwd_bwd_ms": round(fwd_bwd_ms, 4),
                              "fwd_bwd_survey_bytes": fwd_bwd_bytes,
                              "fwd_bwd_frac_of_hbm_peak": round(fwd_bwd_bytes / (fwd_bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        out["valu_roofline"] = valu_utilisation(kern_ms, N, W, H, deg)
        if not args.no_calibration:
            try:
                out["calibration"] = calibration(dev)
            except Exception as ex:  # noqa: BLE001 -- never fail the headline on it
                out["calibration"] = {"error": repr(ex)[:300]}
        if dp.world > 1:
            out["dp_exchange"] = {"form": ("all-gather colour gradients (pipelined with K8 / SH Adam over %d row chunks) + "
                                           "all-reduce geometry gradients" % len(model._color_rows))
                                  if model._color_all is not None else "all-reduce flat gradient buffer",
                                  "payload_bytes_per_rank_per_step": int(dp.bytes_per_step),
                                  "dense_all_reduce_bytes": int(model.params.grad.numel() * 4)}
            out["dp_exchange"]["replicas_identical"] = replicas_identical
            out["dp_exchange"]["transport"] = dp.transport
            if dp.peer is not None:   # which kind of device memory the receive slots live in (tgs_peer_alloc never downgrades silently)
                out["dp_exchange"]["memory_kind"] = dp.peer.memory_kind
            if comm:
                out["dp_exchange"].update(comm)
            out["dp_exchange"]["busbw_sweep"] = busbw
            if transports is not None:
                out["dp_exchange"]["transports"] = transports
        if dp.world == 1 and not args.no_densify_run:
            try:
                del sp, partials, rgb, dacc
                torch.cuda.empty_cache()
                out["train_densify"] = densify_run(N, W, H, deg, args.seed, dev, views, clustered=clustered,
                                                   steps=min(600, max(args.steps * 3, 150)))
            except Exception as ex:  # noqa: BLE001 -- a secondary measurement never fails the headline line
                out["train_densify"] = {"value": None, "error": repr(ex)}
            if args.touch_scene_run:
                try:   # a 720p scene of random Gaussians that starts from touch seeds and grows (throughput of the machinery)
                    torch.cuda.empty_cache()
                    out["train_touch_scene"] = touch_scene_run(dev)[0]
                except Exception as ex:  # noqa: BLE001
                    out["train_touch_scene"] = {"value": None, "error": repr(ex)}
            if args.train_quality != "off":
                try:   # the reference's regime end to end: does the trainer reach a usable image / better depth with touch?
                    torch.cuda.empty_cache()
                    out["train_quality"] = train_quality_run(full=args.train_quality == "full")
                except Exception as ex:  # noqa: BLE001
                    out["train_quality"] = {"value": None, "error": repr(ex)}
        if dp.world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, W, H, deg, args.seed, clustered=clustered)
                # the CPU port times forward+backward only (no SSIM / Adam): the GPU figure in the SAME unit
                out["cpu_baseline"]["gpu_same_unit"] = {"value": round(1e3 / fwd_bwd_ms, 2),
                                                        "unit": out["cpu_baseline"]["unit"]}
            except Exception as ex:  # the oracle is test infrastructure; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
            try:
                # (named for what it is: the default is a quarter-frame sample whose rate is extrapolated; the whole frame
                # takes ~3 min of host time and runs under --cpu-baseline-torch-full -- VERDICT r5 weak #8)
                out["cpu_baseline"]["torch" if args.cpu_baseline_torch_full else "torch_quarter_frame_extrapolated"] = \
                    cpu_baseline_torch("cfg2", full=args.cpu_baseline_torch_full)
            except Exception as ex:  # noqa: BLE001
                out["cpu_baseline"]["torch_error"] = repr(ex)[:300]
    dp.barrier()
    if dp.peer is not None:      # peer transport: unmap the other ranks' buffers before the group goes away
        dp.check_transport()
        dp.peer.close()
        dp.peer = None
    if dp.active:
        import torch.distributed as dist
        dist.destroy_process_group()
    if out is not None:
        _emit(out)    # the only thing on stdout


if __name__ == "__main__":
    main()
