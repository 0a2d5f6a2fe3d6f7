"""Shared helpers for the parity tests (oracle side lives in oracle/, product in touch_gs_amd/)."""
import numpy as np
import torch

from oracle import torch_oracle as O


def scene(N, W, H, deg, seed, view=1, nviews=8, bg=(0.1, 0.2, 0.3)):
    """Seeded synthetic scene -> (fp64 param dict, oracle Camera)."""
    P, c = O.synthetic_scene(N, W, H, deg, seed)
    cam = O.Camera(viewmat=O.orbit_viewmat(view, nviews), **c, bg=bg)
    return P, cam


def clamp_scene(N, W, H, deg, seed, frac=3, logit=12.0, grow=2.0, **kw):
    """Scene that exercises the alpha = 0.999 clamp of App. B.6/B.7: every ``frac``-th Gaussian
    gets opacity logit ``logit`` (sigmoid > 0.999) and ``exp(grow)`` times larger axes, so that many
    (pixel, Gaussian) pairs sit on the clamp."""
    P, cam = scene(N, W, H, deg, seed, **kw)
    P["opac_logit"][::frac] = logit
    P["log_scales"][::frac] += grow
    return P, cam


def count_clamped(pr, gid, ts, cam):
    """Number of (pixel, Gaussian) pairs with o*exp(-sigma) > 0.999 among the listed pairs."""
    import numpy as np
    TW, TH = cam.tiles
    n = 0
    xy, conic, opac = pr["xy"].detach(), pr["conic"].detach(), pr["opac"].detach()
    for t in range(TW * TH):
        s, e = int(ts[t]), int(ts[t + 1])
        if e <= s:
            continue
        g = torch.from_numpy(np.asarray(gid[s:e]).astype(np.int64))
        ty, tx = divmod(t, TW)
        ys = torch.arange(ty * 16, min(ty * 16 + 16, cam.H), dtype=xy.dtype) + cam.pix_center
        xs = torch.arange(tx * 16, min(tx * 16 + 16, cam.W), dtype=xy.dtype) + cam.pix_center
        dx = xy[g, 0][:, None, None] - xs[None, None, :]
        dy = xy[g, 1][:, None, None] - ys[None, :, None]
        a, b, c = (conic[g, i][:, None, None] for i in range(3))
        araw = opac[g][:, None, None] * torch.exp(-(0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy))
        n += int((araw > 0.999).sum())
    return n


def to_dev(P, dev):
    return {k: v.detach().to(torch.float32).to(dev).contiguous() for k, v in P.items()}


def amd_cam(cam):
    from touch_gs_amd import Camera
    return Camera(cam.viewmat.numpy(), cam.fx, cam.fy, cam.cx, cam.cy, cam.W, cam.H, cam.near,
                  cam.pix_center, cam.bg)


def relerr(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / (np.abs(b) + floor)


def splat_fields(splats, radii=None):
    """HIP splat records [N,12] -> dict of fp64 tensors + the packed tile rect (x0,y0,x1,y1)."""
    s = splats.detach().cpu()
    r = s[:, 10].contiguous().view(torch.int32).long() & 0xFFFFFFFF
    x0, y0, w, h = r & 255, (r >> 8) & 255, (r >> 16) & 255, (r >> 24) & 255
    # slots 0, 1 = screen position relative to the rect origin (16 x0, 16 y0): include/tgs.h
    xy = s[:, 0:2].double() + 16.0 * torch.stack([x0, y0], 1).double()
    out = dict(xy=xy, xy_rel=s[:, 0:2].double(), depth=s[:, 2].double(), opac=s[:, 3].double(),
               conic=s[:, 4:7].double(), rgb=s[:, 7:10].double(),
               rect=torch.stack([x0, y0, x0 + w, y0 + h], 1), hits=w * h,
               visible=s[:, 4] > 0)
    if radii is not None:
        out["radius"] = radii.detach().cpu().long()
    return out


def rect_from(xy, radius, cam):
    """App. B.4 on given (fp32-valued) xy / radius, evaluated exactly like the kernel (fp32)."""
    TW, TH = cam.tiles
    u = xy[:, 0].float()
    v = xy[:, 1].float()
    r = radius.float()
    x0 = torch.clamp(((u - r) / 16).to(torch.int64), 0, TW)
    x1 = torch.clamp(((u + r) / 16).to(torch.int64) + 1, 0, TW)
    y0 = torch.clamp(((v - r) / 16).to(torch.int64), 0, TH)
    y1 = torch.clamp(((v + r) / 16).to(torch.int64) + 1, 0, TH)
    return torch.stack([x0, y0, x1, y1], 1)


# ---------------------------------------------------------------------------------------------
# condition-aware gradient comparison (full-size oracle tests)
# ---------------------------------------------------------------------------------------------
K7_KEYS = ("v_xy", "v_conic", "v_opac", "v_rgb", "v_depth")
PARAM_KEYS = ("v_means", "v_log_scales", "v_quats", "v_opac_logit", "v_sh")


def k7_outputs(v_splats):
    """tgs_reduce_partials record [N,12] -> the oracle's blend_bwd keys (fp64, [N,c])."""
    v = np.asarray(v_splats, np.float64)
    return dict(v_xy=v[:, 0:2], v_depth=v[:, 2:3], v_opac=v[:, 3:4], v_conic=v[:, 4:7], v_rgb=v[:, 7:10])


def param_mass(R, Pn, deg, cb, W, H, radius, m7):
    """|J| m: the un-cancelled magnitude of the five parameter gradients, from the magnitudes ``m7`` of the ten
    screen-space gradients (oracle blend_bwd(mass=True)).  The projection backward is linear in its inputs and
    Gaussians are independent, so column c of every Gaussian's Jacobian is the oracle's project_bwd of the c-th
    unit vector: ten calls.  A sum evaluated in floating point is accurate relative to THIS quantity -- the sum
    of the magnitudes of its terms -- not relative to a result that happens to cancel."""
    N = Pn["means"].shape[0]
    shapes = dict(v_xy=2, v_conic=3, v_opac=1, v_rgb=3, v_depth=1)
    acc = None
    for key in K7_KEYS:
        for c in range(shapes[key]):
            unit = {k: np.zeros((N, n) if n > 1 else N) for k, n in shapes.items()}
            if shapes[key] > 1:
                unit[key][:, c] = 1.0
            else:
                unit[key][:] = 1.0
            col = R.project_bwd(Pn["means"], Pn["log_scales"], Pn["quats"], Pn["opac_logit"], Pn["sh"], deg, cb, W, H,
                                radius, unit["v_xy"], unit["v_conic"], unit["v_opac"], unit["v_rgb"], unit["v_depth"])
            w = np.asarray(m7[key], np.float64).reshape(N, -1)[:, c]
            if acc is None:
                acc = {k: np.zeros_like(col[k].reshape(N, -1)) for k in PARAM_KEYS}
            for k in PARAM_KEYS:
                acc[k] += np.abs(col[k].reshape(N, -1)) * w[:, None]
    return acc


def err_over_mass(got, ref, mass):
    """Per Gaussian: max_c |got - ref| / max_c mass  (components of one gradient share their unit, and the kernels
    form them from shared sums, so the Gaussian's largest component magnitude is the scale of all of them)."""
    got, ref, mass = (np.asarray(a, np.float64).reshape(np.asarray(ref).shape[0], -1) for a in (got, ref, mass))
    return np.abs(got - ref).max(1) / (mass.max(1) + 1e-300)


def free_port() -> str:
    """A TCP port nobody listens on right now (for torch.distributed.run --master-port): fixed numbers collide when two
    multi-process tests run back to back and the first one's listener is still closing."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])
