"""CPU tests: the C-ABI library loads and exports every symbol include/tgs.h declares; argument
validation (no GPU work is launched); host-side logic (layouts, split rule, cameras, DP)."""
import ctypes as C
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from tests.util import free_port
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tgs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tgs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from touch_gs_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/tgs.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in touch_gs_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.tgs_version() == 310


def _declared_prototypes():
    """name -> list of parameter kinds ('ptr' | 'int' | 'int64' | 'float' | 'size') parsed from the header."""
    txt = open(os.path.join(ROOT, "include", "tgs.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int64_t|int|size_t|const char\s*\*|void)\s+(tgs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), " ".join(m.group(2).split())
        kinds = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds.append("ptr")
                elif a.startswith("int64_t"):
                    kinds.append("int64")
                elif a.startswith("float"):
                    kinds.append("float")
                elif a.startswith("size_t"):
                    kinds.append("size")
                elif a.startswith("int"):
                    kinds.append("int")
                else:
                    raise AssertionError(f"unparsed parameter {a!r} of {name}")
        protos[name] = kinds
    return protos


def test_ctypes_signatures_match_the_header_prototypes():
    """Arity and parameter kinds of every ctypes signature equal the C prototype in include/tgs.h
    (a drifted signature would pass garbage through the ABI without any error)."""
    from touch_gs_amd import _lib
    protos = _declared_prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    def kind(t):
        if t in (C.c_int, C.c_int32):
            return "int"
        if t is C.c_int64:
            return "int64"
        if t is C.c_float:
            return "float"
        if t is C.c_size_t:
            return "size"
        return "ptr"          # c_void_p, c_char_p, POINTER(...)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        assert [kind(t) for t in argtypes] == protos[name], (name, [kind(t) for t in argtypes], protos[name])


def test_struct_layouts_match_header():
    from touch_gs_amd import _lib
    assert C.sizeof(_lib.TgsCamera) == 16 * 4 + 4 * 4 + 2 * 4 + 2 * 4 + 3 * 4 + 4
    assert C.sizeof(_lib.TgsAdamSpec) == 11 * 4 + 4 + 8   # 11 floats, pad, device pointer
    assert C.sizeof(_lib.TgsLossSpec) == 3 * 8 + 4 * 4
    assert C.sizeof(_lib.TgsRasterOpts) == 6 * 4
    # field order of the header (a per-call option read from the wrong slot would silently select another kernel form)
    txt = open(os.path.join(ROOT, "include", "tgs.h")).read()
    body = re.search(r"typedef struct TgsRasterOpts \{(.*?)\} TgsRasterOpts;", txt, flags=re.S).group(1)
    assert re.findall(r"int32_t\s+(\w+);", body) == [f[0] for f in _lib.TgsRasterOpts._fields_]


def test_argument_validation_without_gpu():
    from touch_gs_amd import _lib
    lib = _lib.load()
    assert lib.tgs_num_groups(1000) == 4 and lib.tgs_num_groups(0) == 0
    assert lib.tgs_num_tiles(1920, 1080) == 120 * 68
    assert lib.tgs_sort_scratch_bytes(1000) >= 28 * 1000
    cam = _lib.TgsCamera()  # W = H = 0 -> invalid
    rc = lib.tgs_project_fwd(C.byref(cam), 10, None, None, None, None, None, 0, -1, None, None, None, None)
    assert rc == -1 and b"camera" in lib.tgs_last_error()
    cam.W, cam.H, cam.fx, cam.fy = 64, 64, 50.0, 50.0
    rc = lib.tgs_project_fwd(C.byref(cam), 10, None, None, None, None, None, 0, -1, None, None, None, None)
    assert rc == -1 and b"null" in lib.tgs_last_error()
    assert lib.tgs_project_fwd(C.byref(cam), 0, None, None, None, None, None, 0, -1, None, None, None, None) == 0
    with pytest.raises(RuntimeError):
        _lib.check(rc, "x")
    with pytest.raises(RuntimeError):  # CPU tensors are rejected: there is no CPU path
        _lib.ptr(torch.zeros(4))


def test_ops_refuse_cpu_tensors():
    from touch_gs_amd import Camera, ops
    cam = Camera(np.eye(4), 50, 50, 32, 32, 64, 64)
    z = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        ops.project_fwd(cam, z, z, torch.zeros(4, 4), torch.zeros(4), None, -1)


def test_param_layout_alignment():
    from touch_gs_amd.optim import GaussianParams, layout
    for N in (1, 2, 3, 5, 1000, 1001):
        L = layout(N, 16)
        for k in GaussianParams.NAMES:
            assert L[k][0] % 4 == 0
        assert L["total"] % 4 == 0 and L["total"] >= 59 * N
    gp = GaussianParams.allocate(5, 4, "cpu")
    gp.quats.fill_(2.0)
    assert float(gp.flat.sum()) == 40.0
    assert gp.g["sh"].shape == (5, 4, 3)


def test_split_rule_matches_reference_goldens():
    """Goldens captured from the reference's utils/create_point_cloud_from_touches.py:174-198
    (SURVEY 8b): n=151,f=0.08 -> 13 idx starting [0,11,23,34,46]; n=100,f=0.13 -> 13; n=40,f=0.8 -> 32."""
    from touch_gs_amd.parallel import split_train_indices
    tr, ev = split_train_indices(151, 0.08)
    assert len(tr) == 13 and tr[:5] == [0, 11, 23, 34, 46] and len(ev) == 138
    assert len(split_train_indices(100, 0.13)[0]) == 13
    assert len(split_train_indices(40, 0.8)[0]) == 32


def test_camera_opengl_conversion():
    from touch_gs_amd import Camera
    c2w = np.eye(4)
    c2w[:3, 3] = [1.0, 2.0, 3.0]
    cam = Camera.from_c2w_opengl(c2w, 100, 100, 50, 50, 100, 100)
    # a point one unit in front of an OpenGL camera (-z) has OpenCV depth +1
    p = np.array([1.0, 2.0, 2.0, 1.0])
    assert np.allclose(cam.viewmat @ p, [0, 0, 1, 1])
    cs = cam.c_struct()
    assert cs.W == 100 and abs(cs.viewmat[11] - cam.viewmat[2, 3]) < 1e-6


def test_scene_generator_matches_oracle_recipe():
    from oracle import torch_oracle as O
    from touch_gs_amd.scene import orbit_viewmat, synthetic_gaussians
    P, intr = synthetic_gaussians(500, 160, 96, 3, 1236)
    Po, c = O.synthetic_scene(500, 160, 96, 3, 1236)
    for k in P:
        assert torch.allclose(P[k].double(), Po[k], atol=1e-6), k
    assert abs(intr["fx"] - c["fx"]) < 1e-9
    assert np.allclose(orbit_viewmat(3, 8), O.orbit_viewmat(3, 8).numpy())


def test_chunk_ranges():
    from touch_gs_amd.parallel import GradSync
    for numel, n in ((59_000_000, 8), (1003, 5), (8, 8), (4, 3), (12, 1)):
        r = GradSync.chunk_ranges(numel, n)
        assert r[0][0] == 0 and r[-1][1] == numel and all(b % 4 == 0 for b, _ in r)
        assert all(r[i][1] == r[i + 1][0] for i in range(len(r) - 1)) and len(r) <= n


def test_intersect_budget_logic():
    from touch_gs_amd.ops import IntersectBudget
    b = IntersectBudget()
    assert b.initial(1000) == max(8 * 1000, 1 << 16)
    b = IntersectBudget(capacity=123, sync=False)
    assert b.initial(10) == 123 and b.check() is None
    b.last_status = torch.tensor([500, 1], dtype=torch.int32)
    with pytest.raises(RuntimeError):
        b.check()


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from touch_gs_amd import parallel
dp = parallel.init_from_env(backend="gloo")
assert dist.is_initialized() and dp.world == 2
torch.manual_seed(0)
params = torch.randn(1000)                       # identical replicas
m = torch.zeros(1000); v = torch.zeros(1000)
seen = []
for step in range(4):
    view = dp.views_for_step(step, 8)
    seen.append(view)
    g = torch.full((1000,), float(view + 1))      # "gradient of view"
    scale = dp.all_reduce_(g)
    g = g * scale
    expect = (2 * step * 2 + 1 + 2) / 2.0         # mean of (2s+1) and (2s+2)
    assert torch.allclose(g, torch.full((1000,), expect)), (g[0].item(), expect)
    m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
    params = params - 1e-3 * m / (v.sqrt() + 1e-15)
    dp.assert_replicas_identical(params)
assert seen == [(s * 2 + dp.rank) % 8 for s in range(4)]
# pipelined form: chunked all-reduce, optimizer callback per reduced chunk
gflat = torch.arange(1003, dtype=torch.float32) * (dp.rank + 1)
calls = []
def step_range(b, e, scale):
    calls.append((b, e))
    gflat[b:e] *= scale
began = []
dp.n_chunks = 5
dp.reduce_and_step(gflat, step_range, lambda: began.append(1))
assert began == [1] and calls[0][0] == 0 and calls[-1][1] == 1003 and all(b % 4 == 0 for b, _ in calls)
assert all(calls[i][1] == calls[i + 1][0] for i in range(len(calls) - 1)) and len(calls) == 5
assert torch.allclose(gflat, torch.arange(1003, dtype=torch.float32) * 1.5)
assert dp.max_over_ranks(float(dp.rank)) == 1.0
bytes_dense = dp.bytes_per_step
# factored form: all-gather of the colour-gradient blocks, all-reduce of the geometry gradients
Ng = 37
block = torch.full((3 * Ng + 4,), float(dp.rank + 1)); block[3 * Ng:] = torch.tensor([10. * dp.rank, 1., 2., 0.])
allc = torch.zeros(2, 3 * Ng + 4)
geom = torch.arange(44, dtype=torch.float32) * (dp.rank + 1)
order = []
def step_sh(c, scale):
    order.append("sh")
    assert scale == 0.5 and c.shape == (2, 3 * Ng + 4)
    assert torch.all(c[0, :3 * Ng] == 1) and torch.all(c[1, :3 * Ng] == 2)          # rank-major blocks
    assert c[0, 3 * Ng].item() == 0.0 and c[1, 3 * Ng].item() == 10.0              # camera trailers
def step_geom(b, e, scale):
    order.append("geom")
    assert (b, e, scale) == (0, 44, 0.5)
    assert torch.allclose(geom, torch.arange(44, dtype=torch.float32) * 3.0)
dp.gather_color_reduce_geom_and_step(geom, block, allc, step_sh, step_geom, lambda: order.append("begin"))
assert order == ["begin", "sh", "geom"] and dp.bytes_per_step == 4 * (2 * (3 * Ng + 4) + 44)
# pipelined factored form: K8 / gather / SH-Adam per row chunk, then the geometry all-reduce
dp.color_chunks = 3
rows = dp.color_chunk_rows(1000)
assert rows == [(0, 512), (512, 1000)] and dp.color_chunk_rows(100) == [(0, 100)]      # boundaries on 256 rows
dp.color_chunks = 4
assert dp.color_chunk_rows(1024) == [(0, 256), (256, 512), (512, 768), (768, 1024)]
rows = [(0, 512), (512, 1000)]
blocks = [torch.zeros(3 * (e - b) + 4) for b, e in rows]
blocks_all = [torch.zeros(2, 3 * (e - b) + 4) for b, e in rows]
geom2 = torch.zeros(44)
order = []
def backward_chunk(c):
    order.append(("k8", c))
    b, e = rows[c]
    blocks[c][:3 * (e - b)] = torch.arange(3 * b, 3 * e, dtype=torch.float32) + 1000.0 * dp.rank
    blocks[c][3 * (e - b):] = torch.tensor([10. * dp.rank, 1., 2., 0.])
    geom2[22 * c:22 * (c + 1)] = float((c + 1) * (dp.rank + 1))
def step_sh_chunk(c, allc, scale):
    order.append(("sh", c))
    b, e = rows[c]
    assert scale == 0.5 and allc.shape == (2, 3 * (e - b) + 4)
    for r in range(2):
        assert torch.equal(allc[r, :3 * (e - b)], torch.arange(3 * b, 3 * e, dtype=torch.float32) + 1000.0 * r)
        assert allc[r, 3 * (e - b)].item() == 10.0 * r
def step_geom2(b, e, scale):
    order.append("geom")
    assert torch.allclose(geom2, torch.cat([torch.full((22,), 3.0), torch.full((22,), 6.0)]))
dp.pipelined_color_exchange_and_step(geom2, blocks, blocks_all, backward_chunk, step_sh_chunk, step_geom2,
                                     lambda: order.append("begin"))
assert order == ["begin", ("k8", 0), ("k8", 1), ("sh", 0), ("sh", 1), "geom"], order
assert dp.bytes_per_step == 4 * (2 * (3 * 1000 + 8) + 44)
dp.barrier()
if dp.rank == 0: print("GLOO_OK", bytes_dense)
'''


def test_data_parallel_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), ROOT],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GLOO_OK 4012" in r.stdout  # bytes of the last exchanged buffer (1003 floats)


def test_position_lr_schedule():
    """Splatfacto's exponential decay of the position learning rate: 1.6e-4 at step
    0, 1.6e-6 at max_steps and after, geometric in between; None keeps it constant."""
    from touch_gs_amd.model import ModelConfig
    c = ModelConfig()
    assert abs(c.lr_means_at(0) - 1.6e-4) < 1e-12 and abs(c.lr_means_at(30000) - 1.6e-6) < 1e-12
    assert abs(c.lr_means_at(15000) - 1.6e-5) < 1e-10 and c.lr_means_at(10 ** 6) == c.lr_means_at(30000)
    assert ModelConfig(lr_means_final=None).lr_means_at(12345) == 1.6e-4


def test_densify_screen_size_rules_cpu():
    """Refinement decisions on a hand-made population (CPU tensors; refine() is plain torch):
    hot + large screen radius -> split only before stop_screen_size_at; "too big" culls (scale or
    screen size) only after the first opacity-reset interval."""
    from touch_gs_amd.densify import DensifyConfig, DensityController
    from touch_gs_amd.optim import FusedAdam, GaussianParams
    N, K = 8, 1
    def population():
        gp = GaussianParams.allocate(N, K, "cpu")
        gp.quats[:, 0] = 1.0
        gp.log_scales[:] = math.log(0.001)           # small in world units
        gp.opac_logit[:] = 2.0
        gp.opac_logit[7] = -8.0                      # transparent -> always culled
        gp.log_scales[6] = math.log(0.8)             # above cull_scale_thresh
        return gp, FusedAdam(gp, dict(means=1e-4, log_scales=1e-3, quats=1e-3, opac_logit=1e-2, sh_dc=1e-3, sh_rest=1e-4))
    cfg = DensifyConfig(warmup_length=0, refine_every=10, reset_alpha_every=3, stop_screen_size_at=100)
    def stats(dc):
        dc.vis_count[:] = 1.0
        dc.grad_norm_sum[:] = 0.0
        dc.grad_norm_sum[[0, 1, 2]] = 1.0            # hot: 0 (small), 1 (large on screen), 2 (very large on screen)
        dc.max_radius[1] = 0.06                      # > split_screen_size
        dc.max_radius[2] = 0.2                       # > cull_screen_size
        dc.max_radius[3] = 0.2                       # cold but huge on screen
    # (a) early (step 20 <= refine_every * reset_alpha_every = 30): no "too big" culls yet
    gp, opt = population()
    dc = DensityController(cfg, N, "cpu"); stats(dc)
    _, _, info = dc.refine(gp, opt, 20)
    assert info["culled"] == 1 and info["cloned"] == 1 and info["split"] == 2      # 0 cloned; 1, 2 split (screen size)
    assert info["after"] == N - 1 - 2 + 1 + 2 * 2
    # (b) after the first reset interval, still in the screen-size phase: scale and screen-size culls
    gp, opt = population()
    dc = DensityController(cfg, N, "cpu"); stats(dc)
    _, _, info = dc.refine(gp, opt, 50)
    assert info["culled"] == 4 and info["cloned"] == 1 and info["split"] == 1      # culled: 7, 6, 2, 3; split: 1
    # (c) past stop_screen_size_at: the screen radius no longer matters
    gp, opt = population()
    dc = DensityController(cfg, N, "cpu"); stats(dc)
    _, _, info = dc.refine(gp, opt, 140)
    assert info["culled"] == 2 and info["cloned"] == 3 and info["split"] == 0      # culled: 7, 6; clones: 0, 1, 2
    # (d) Splatfacto's pause after an opacity reset: at step % reset_interval == refine_every the opacities are reset and
    # nothing is split or culled until num_train_data + refine_every steps have passed (every image seen again)
    gp, opt = population()
    dc = DensityController(cfg, N, "cpu"); stats(dc)
    new, _, info = dc.refine(gp, opt, 130)          # 130 % 30 == 10
    assert info["opacity_reset"] and info["culled"] == info["cloned"] == info["split"] == 0 and info["after"] == N
    assert float(torch.sigmoid(new.opac_logit).max()) <= 2 * cfg.cull_alpha_thresh + 1e-6
    import dataclasses as _dc
    dc = DensityController(_dc.replace(cfg, num_train_data=15), N, "cpu"); stats(dc)
    gp, opt = population()
    _, _, info = dc.refine(gp, opt, 140)            # 140 % 30 = 20 <= 15 + 10: still paused
    assert not info["opacity_reset"] and info["culled"] == info["cloned"] == info["split"] == 0
    # (e) after stop_split_at: culls only (continue_cull_post_densification), and no more opacity resets
    dc = DensityController(_dc.replace(cfg, stop_split_at=100), N, "cpu"); stats(dc)
    gp, opt = population()
    _, _, info = dc.refine(gp, opt, 130)
    assert not info["opacity_reset"] and info["culled"] == 2 and info["cloned"] == info["split"] == 0


def test_unseen_cull_needs_a_complete_window_cpu():
    """DensifyConfig.cull_unseen (off by default: not in Splatfacto) removes Gaussians no training view has had in its
    frustum -- but only at refinements that cull at all (not in the pause after an opacity reset) and only when the
    window is KNOWN to have shown every training view: ``num_train_data`` different view identities passed to
    ``accumulate``, or, for callers that pass none (nerfstudio's datamanager shuffles per epoch), a window of at least
    2 n - 1 steps (``refine_every >= n`` only covers every view for a round-robin order)."""
    import dataclasses as _dc
    from touch_gs_amd.densify import DensifyConfig, DensityController
    from touch_gs_amd.optim import FusedAdam, GaussianParams
    N, n_views = 6, 4
    def population():
        gp = GaussianParams.allocate(N, 1, "cpu")
        gp.quats[:, 0] = 1.0
        gp.log_scales[:] = math.log(0.001)
        gp.opac_logit[:] = 2.0
        return gp, FusedAdam(gp, dict(means=1e-4, log_scales=1e-3, quats=1e-3, opac_logit=1e-2, sh_dc=1e-3, sh_rest=1e-4))
    radii = torch.tensor([3, 3, 3, 3, 0, 0], dtype=torch.int32)       # rows 4, 5 are outside every frustum
    def run(cfg, keys, step=20):
        gp, opt = population()
        dc = DensityController(cfg, N, "cpu")
        for k in keys:
            dc.accumulate(torch.zeros(N, 2), radii, 64, 64, view_key=k)
        return dc.refine(gp, opt, step)[2]["culled"]
    base = DensifyConfig(warmup_length=0, refine_every=10, reset_alpha_every=0, num_train_data=n_views)
    assert DensifyConfig().cull_unseen is False and run(base, [0, 1, 2, 3]) == 0       # the default never does it
    on = _dc.replace(base, cull_unseen=True)
    assert run(on, [0, 1, 2, 3]) == 2                                                   # every view seen: 4, 5 go
    assert run(on, [0, 1, 2, 2, 1, 0, 1, 2, 0, 1]) == 0                                 # ten steps, view 3 never shown
    assert run(on, [None] * 10) == 2                                                    # anonymous, 10 >= 2 * 4 - 1
    assert run(_dc.replace(on, refine_every=6), [None] * 6, step=18) == 0               # anonymous, 6 < 7: could have missed one
    paused = _dc.replace(on, reset_alpha_every=3)                                       # reset interval 30: steps 30 .. 44 pause
    assert run(paused, [0, 1, 2, 3], step=40) == 0 and run(paused, [0, 1, 2, 3], step=50) == 2


def test_refine_places_children_behind_their_parents_cpu():
    """DensityController.refine builds the refined store with one gather: survivors keep their relative
    order (values and Adam moments bit for bit), a clone sits directly behind its source, the
    n_split_samples replacements of a split parent occupy its place, new rows start with zero moments,
    and the split samples are mean + R (noise * scale) with scale / 1.6 for the documented noise
    assignment -- so a Morton-ordered buffer stays ordered without the argsort the trainer used to run."""
    from touch_gs_amd.densify import DensifyConfig, DensityController, quat_to_rotmat
    from touch_gs_amd.optim import FusedAdam, GaussianParams
    g = torch.Generator().manual_seed(3)
    N, K = 600, 4
    vals = dict(means=torch.randn(N, 3, generator=g), log_scales=torch.randn(N, 3, generator=g) * 0.5 - 4,
                quats=torch.randn(N, 4, generator=g), opac_logit=torch.randn(N, generator=g) * 2,
                sh=torch.randn(N, K, 3, generator=g))
    gp = GaussianParams.from_tensors(*[vals[k] for k in GaussianParams.NAMES])
    opt = FusedAdam(gp, dict(means=1e-4, log_scales=1e-3, quats=1e-3, opac_logit=1e-2, sh_dc=1e-3, sh_rest=1e-4))
    opt.exp_avg.copy_(torch.randn(gp.flat.shape, generator=g))
    opt.exp_avg_sq.copy_(torch.rand(gp.flat.shape, generator=g))
    opt.t = 7
    cfg = DensifyConfig(warmup_length=0, refine_every=10, densify_grad_thresh=0.5, densify_size_thresh=0.02,
                        reset_alpha_every=0, n_split_samples=2)
    dc = DensityController(cfg, N, "cpu")
    dc.grad_norm_sum = torch.rand(N, generator=g) * 2
    dc.vis_count = torch.ones(N)
    step = 10
    hot = dc.grad_norm_sum > cfg.densify_grad_thresh
    big = torch.exp(gp.log_scales).max(-1).values > cfg.densify_size_thresh
    cull = torch.sigmoid(gp.opac_logit) < cfg.cull_alpha_thresh
    split, clone = hot & big & ~cull, hot & ~big & ~cull
    keep = ~cull & ~split
    old = {k: getattr(gp, k).clone() for k in GaussianParams.NAMES}
    m_old = {k: v.clone() for k, v in GaussianParams.views_of(opt.exp_avg, N, K).items()}
    new_p, new_o, info = dc.refine(gp, opt, step)
    S = cfg.n_split_samples
    assert info["cloned"] == int(clone.sum()) and info["split"] == int(split.sum()) and info["culled"] == int(cull.sum())
    assert new_p.N == int(keep.sum() + clone.sum() + S * split.sum()) and new_o.t == 7
    m_new = GaussianParams.views_of(new_o.exp_avg, new_p.N, K)
    noise = torch.randn(S * int(split.sum()), 3, generator=torch.Generator().manual_seed(1_000_003 * (step + 1)))
    pos, j_split = 0, 0
    for i in range(N):
        if keep[i]:
            for k in GaussianParams.NAMES:
                assert torch.equal(getattr(new_p, k)[pos], old[k][i]) and torch.equal(m_new[k][pos], m_old[k][i])
            pos += 1
            if clone[i]:
                for k in GaussianParams.NAMES:
                    assert torch.equal(getattr(new_p, k)[pos], old[k][i]) and not m_new[k][pos].any()
                pos += 1
        elif split[i]:
            R = quat_to_rotmat(old["quats"][i:i + 1])[0]
            for s_ in range(S):
                nz = noise[s_ * int(split.sum()) + j_split]
                want = old["means"][i] + R @ (nz * torch.exp(old["log_scales"][i]))
                assert torch.allclose(new_p.means[pos], want, atol=1e-6)
                assert torch.allclose(new_p.log_scales[pos], old["log_scales"][i] - math.log(1.6), atol=1e-6)
                assert torch.equal(new_p.quats[pos], old["quats"][i]) and not m_new["means"][pos].any()
                pos += 1
            j_split += 1
    assert pos == new_p.N
    # the statistics restart at the new size
    assert dc.vis_count.shape[0] == new_p.N and not dc.vis_count.any()
    # an overflowed frame (guard[1] != 0) contributes nothing to the statistics
    dc.accumulate(torch.ones(new_p.N, 2), torch.ones(new_p.N, dtype=torch.int32), 64, 64, guard=torch.tensor([5, 1], dtype=torch.int32))
    assert not dc.vis_count.any() and not dc.grad_norm_sum.any() and not dc.max_radius.any()
    dc.accumulate(torch.ones(new_p.N, 2), torch.ones(new_p.N, dtype=torch.int32), 64, 64, guard=torch.tensor([5, 0], dtype=torch.int32))
    assert dc.vis_count.sum() == new_p.N


def test_balanced_order_deals_long_runs_over_the_groups_cpu():
    """optim.balanced_order: a permutation; rows whose `hits` exceed 32 sit at the END of the 256-row groups, dealt by size
    rank round robin (every group gets the same number +- 1 and a similar sum); the other rows keep their Morton order
    among themselves; without hits (or without long runs) it IS the Morton order."""
    from touch_gs_amd.optim import balanced_order, morton_order
    g = torch.Generator().manual_seed(5)
    N = 256 * 37 + 91
    means = torch.randn(N, 3, generator=g)
    hits = (torch.rand(N, generator=g) ** 10 * 3600).long()              # ~7 % above 32, a heavy tail
    assert torch.equal(balanced_order(means, None), morton_order(means))
    assert torch.equal(balanced_order(means, torch.zeros(N, dtype=torch.long)), morton_order(means))
    perm = balanced_order(means, hits)
    assert torch.equal(torch.sort(perm).values, torch.arange(N))
    h = hits[perm]
    big = h > 32
    G = N // 256
    per_group = big[:G * 256].view(G, 256)
    cnt = per_group.sum(1)
    assert int(cnt.max()) - int(cnt.min()) <= 1 and not big[G * 256:].any()        # dealt evenly; the partial group takes none
    for gi in range(G):                                                           # ... at the end of every group
        assert per_group[gi, 256 - int(cnt[gi]):].all() and not per_group[gi, :256 - int(cnt[gi])].any()
    sums = torch.where(big, h, torch.zeros_like(h))[:G * 256].view(G, 256).sum(1).float()
    plain = hits[morton_order(means)]
    plain_sums = torch.where(plain > 32, plain, torch.zeros_like(plain))[:G * 256].view(G, 256).sum(1).float()
    assert sums.max() / sums.mean() < 1.25 < plain_sums.max() / plain_sums.mean()   # measured 1.08 against 2.3
    inv = torch.empty(N, dtype=torch.long)
    inv[morton_order(means)] = torch.arange(N)
    small_rows = perm[~big]
    assert (inv[small_rows][1:] > inv[small_rows][:-1]).all()                      # Morton order among the short-run rows


def test_morton_order_and_permute_cpu():
    """optim.morton_order sorts along a 3-D Z-order curve (neighbours in memory are neighbours in space)
    and GaussianParams.permute_ moves every segment of every flat buffer consistently."""
    from touch_gs_amd.optim import FusedAdam, GaussianParams, morton_order
    g = torch.Generator().manual_seed(0)
    N, K = 4096, 4
    gp = GaussianParams.allocate(N, K, "cpu")
    gp.flat.copy_(torch.randn(gp.flat.shape, generator=g))
    opt = FusedAdam(gp, dict(means=1e-4, log_scales=1e-3, quats=1e-3, opac_logit=1e-2, sh_dc=1e-3, sh_rest=1e-4))
    opt.exp_avg.copy_(torch.randn(gp.flat.shape, generator=g))
    before = {k: getattr(gp, k).clone() for k in GaussianParams.NAMES}
    m_before = GaussianParams.views_of(opt.exp_avg.clone(), N, K)
    perm = morton_order(gp.means)
    assert sorted(perm.tolist()) == list(range(N))
    gp.permute_(perm, opt.exp_avg)
    for k in GaussianParams.NAMES:
        assert torch.equal(getattr(gp, k), before[k][perm])
        assert torch.equal(GaussianParams.views_of(opt.exp_avg, N, K)[k], m_before[k][perm])
    d_sorted = (gp.means[1:] - gp.means[:-1]).norm(dim=1).median()
    d_before = (before["means"][1:] - before["means"][:-1]).norm(dim=1).median()
    assert d_sorted < 0.35 * d_before
    # identical points, degenerate extents: still a permutation
    same = torch.zeros(7, 3)
    assert sorted(morton_order(same).tolist()) == list(range(7))


def test_bench_byte_models_match_survey():
    """bench.survey_bytes is SURVEY 8(d)'s compulsory-traffic model split per kernel: its forward and
    backward sums must reproduce B_fwd = 312 N + 88 I + 24 P + 16 T and B_bwd = 84 I + 28 P + 512 N
    (SH degree 3), e.g. 785 MB / 974 MB at the survey's cfg3 estimate."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    N, I, P, T, K = 1_000_000, 4_810_000, 1920 * 1080, 8160, 16
    sb = bench.survey_bytes(N, I, P, T, K)
    fwd = sb["project_bin_sort"] + sb["raster_fwd"]
    bwd = sb["raster_bwd"] + sb["project_bwd"]
    assert fwd == 312 * N + 88 * I + 24 * P + 16 * T and abs(fwd / 1e6 - 785) < 1
    assert bwd == 84 * I + 28 * P + 512 * N and abs(bwd / 1e6 - 974) < 1
    assert sb["adam"] == 1652 * N
    assert set(bench.CONFIGS) >= {"cfg2", "cfg3", "cfg5", "clustered"} and bench.CONFIGS["cfg3"]["seed"] == 1236
    ab = bench.algorithmic_bytes(N, I, P, T, K)
    assert set(ab) == set(sb)


def test_xcd_schedule_sizes_and_bands_host():
    """Host-side size functions of the K6 / K7 schedule (no GPU): tgs_tile_order_len = 8 XCDs x the slots of
    one XCD (tiles dealt in granules of 8), and the image bands of tgs_rasterize_bwd_band partition the
    row-major tile range in order, each band <= 8 x 1024 tiles."""
    import ctypes as C
    from touch_gs_amd import _lib
    lib = _lib.load()
    for W, H in [(16, 16), (70, 50), (800, 800), (1920, 1080), (3840, 2160), (4080, 4080)]:
        T = lib.tgs_num_tiles(W, H)
        L = lib.tgs_tile_order_len(W, H)
        per = -(-(-(-T // 8)) // 8) * 8
        assert L == 8 * per and L >= T
        nb = lib.tgs_num_bands(W, H)
        assert nb >= 1
        t0, t1 = C.c_int(), C.c_int()
        nxt = 0
        for b in range(nb):
            assert lib.tgs_band_tiles(W, H, b, C.byref(t0), C.byref(t1)) == 0
            assert t0.value == min(nxt, T) and t0.value <= t1.value <= T and t1.value - t0.value <= 8 * 1024
            nxt = t1.value if t1.value > t0.value else nxt
        assert nxt == T
        assert lib.tgs_band_tiles(W, H, nb, C.byref(t0), C.byref(t1)) != 0


def test_c_abi_refuses_a_short_tile_start_buffer():
    """TGS_VERSION 300: every entry point that takes tile_start takes the length the caller allocated and fails with
    TGS_E_ARG -- before any launch, so this runs without a GPU -- when it is shorter than tgs_tile_start_len(W, H): a
    C caller that still allocates T + 1 ints (ABI <= 200) gets an error instead of out-of-bounds device writes."""
    from touch_gs_amd import _lib
    lib = _lib.load()
    W, H = 320, 208
    T = lib.tgs_num_tiles(W, H)
    need = lib.tgs_tile_start_len(W, H)
    assert need == T + 1 + 512
    cam = _lib.TgsCamera()
    cam.W, cam.H, cam.fx, cam.fy, cam.cx, cam.cy = W, H, 300.0, 300.0, W / 2, H / 2
    for i in (0, 5, 10, 15):
        cam.viewmat[i] = 1.0
    fake = C.c_void_p(0x1000)      # never dereferenced: the length check precedes every launch
    for short in (T + 1, need - 1):
        rc = lib.tgs_rasterize_fwd(C.byref(cam), fake, fake, fake, short, None, fake, fake, fake, None, None, None, None, None)
        assert rc == -1 and b"tgs_tile_start_len" in lib.tgs_last_error(), lib.tgs_last_error()
        rc = lib.tgs_rasterize_bwd(C.byref(cam), fake, fake, fake, fake, short, None, fake, fake, fake, fake, None, None, None,
                                   None, fake, None, None, None, None)
        assert rc == -1 and b"tgs_tile_start_len" in lib.tgs_last_error(), lib.tgs_last_error()
        rc = lib.tgs_bin_sort(C.byref(cam), 0, None, fake, fake, short, fake, fake, None, 1024, fake, fake, None, -1, None)
        assert rc == -1 and b"tgs_tile_start_len" in lib.tgs_last_error(), lib.tgs_last_error()
        rc = lib.tgs_project_bin_sort(C.byref(cam), 0, None, None, None, None, None, 0, -1, None, None, fake, fake, short, fake,
                                      fake, None, 1024, fake, fake, None, -1, None)
        assert rc == -1 and b"tgs_tile_start_len" in lib.tgs_last_error(), lib.tgs_last_error()
        rc = lib.tgs_project_bin_sort_front(C.byref(cam), 256, fake, fake, fake, fake, fake, 16, 3, fake, None, fake, fake, short,
                                            fake, fake, None, 1024, fake, fake, None, -1, fake, 0, None, None, None, None)
        assert rc == -1 and b"tgs_tile_start_len" in lib.tgs_last_error(), lib.tgs_last_error()


def test_raster_defaults_are_process_wide_and_opts_override_them():
    """The tgs_set_* calls only move the process-wide DEFAULTS (atomic words); they round-trip, and a TgsRasterOpts with
    every field -1 is what NULL means."""
    from touch_gs_amd import _lib, ops
    lib = _lib.load()
    before = lib.tgs_set_raster_variant(-1, -1), lib.tgs_set_k6_split(-1), lib.tgs_set_k7_quad(-1, -1)
    try:
        assert lib.tgs_set_raster_variant(0, 1) == 2 and lib.tgs_set_raster_variant(-1, -1) == 2
        assert lib.tgs_set_raster_variant(1, 0) == 1
        assert lib.tgs_set_k6_split(7) == 7 and lib.tgs_set_k6_split(-1) == 7
        assert lib.tgs_set_k7_quad(3, 100) == (3 | 100 << 8) and ops.set_k7_quad() == (3, 100)
    finally:
        lib.tgs_set_raster_variant(before[0] & 1, (before[0] >> 1) & 1)
        lib.tgs_set_k6_split(before[1])
        lib.tgs_set_k7_quad(before[2] & 255, before[2] >> 8)
    o = ops.raster_opts()
    assert [getattr(o, f[0]) for f in o._fields_] == [-1] * 6
    o = ops.raster_opts(k6_blocks=False, k7_quad=0)
    assert (o.k6_blocks, o.k6_split, o.k7_front_to_back, o.k7_quad, o.k7_quad_min_walk) == (0, -1, -1, 0, -1)


def test_tile_start_copies_are_refused():
    """The rasterizer owns 512 scratch ints behind the tile starts; the Python ops refuse a tensor that
    has lost them (a clone of the T + 1 view) with the reason, before the C ABI's own length check would."""
    from touch_gs_amd import ops
    T = 40
    buf = torch.zeros(T + 1 + 512, dtype=torch.int32)
    ops._check_tile_start(buf[:T + 1], T)
    with pytest.raises(ValueError):
        ops._check_tile_start(buf[:T + 1].clone(), T)
    with pytest.raises(ValueError):
        ops._check_tile_start(buf[8:T + 9], T)
