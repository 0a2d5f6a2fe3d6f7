"""Full-size (BASELINE cfg3: 1 M Gaussians, 1920x1080, SH 3) checks through size-independent
properties -- the oracle cannot run at this size in seconds, these invariants can."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N, W, H, DEG, SEED = 1_000_000, 1920, 1080, 3, 1236


@pytest.fixture(scope="module")
def frame(dev):
    from touch_gs_amd import ops
    from touch_gs_amd.scene import make_camera, synthetic_gaussians
    P, intr = synthetic_gaussians(N, W, H, DEG, SEED)
    D = {k: v.to(dev).contiguous() for k, v in P.items()}
    cam = make_camera(intr, 0, 8, bg=(0.1, 0.2, 0.3))
    sp, radii, gb, ts, sg, st = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                     D["sh"], DEG, want_radii=True)
    n = int(ts[-1])                 # total length of the tile lists
    assert n == st.tolist()[0]
    rgb, depth, fT, fidx = ops.rasterize_fwd(cam, sp, sg, ts, want_idx=True)
    return dict(D=D, cam=cam, sp=sp, radii=radii, gb=gb, ts=ts, sg=sg, n=n, n_index=st.tolist()[0], rgb=rgb, depth=depth, fT=fT, fidx=fidx)


def test_tile_lists_sorted_and_complete(frame):
    sp, ts, sg, n = frame["sp"], frame["ts"], frame["sg"][:frame["n"]].long(), frame["n"]
    assert int(ts[-1]) == n and bool((ts[1:] >= ts[:-1]).all())
    # every list is ordered by (depth bits, gid): compare neighbours that are in the same tile
    depth_bits = sp[:, 2].contiguous().view(torch.int32).long()
    key = (depth_bits[sg] << 32) | sg
    tile_of = torch.searchsorted(ts[1:].long().contiguous(), torch.arange(n, device=sg.device), right=True)
    same = tile_of[1:] == tile_of[:-1]
    assert bool((key[1:][same] > key[:-1][same]).all())     # strictly increasing -> also no duplicates
    # the number of pairs equals the sum of the packed rect areas
    r = sp[:, 10].contiguous().view(torch.int32).long() & 0xFFFFFFFF
    assert int((((r >> 16) & 255) * ((r >> 24) & 255)).sum()) == n
    # every listed Gaussian's rect contains the tile it is listed in
    x0, y0, w, h = r & 255, (r >> 8) & 255, (r >> 16) & 255, (r >> 24) & 255
    TW = (W + 15) // 16
    tx, ty = tile_of % TW, tile_of // TW
    g = sg
    assert bool(((tx >= x0[g]) & (tx < x0[g] + w[g]) & (ty >= y0[g]) & (ty < y0[g] + h[g])).all())


def test_compositing_identities_fullsize(frame):
    rgb, depth, fT, fidx = frame["rgb"], frame["depth"], frame["fT"], frame["fidx"]
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    assert float(fT.min()) > 1e-4 - 1e-9 and float(fT.max()) <= 1.0      # a pixel never drops below the stop threshold
    # pixels without any contributor show the pure background and zero depth
    empty = fidx < 0
    if bool(empty.any()):
        bg = torch.tensor(frame["cam"].bg, device=rgb.device)
        assert torch.allclose(rgb[empty], bg.expand(int(empty.sum()), 3)) and float(depth[empty].abs().max()) == 0
        assert float((fT[empty] - 1).abs().max()) == 0
    # depth_acc / alpha lies between the nearest and farthest visible depth
    a = 1 - fT
    z = frame["sp"][:, 2][frame["sp"][:, 4] > 0]
    dhat = depth[a > 1e-3] / a[a > 1e-3]
    assert float(dhat.min()) >= float(z.min()) - 1e-3 and float(dhat.max()) <= float(z.max()) + 1e-3


def test_bitwise_determinism_and_fused_front_half(frame):
    from touch_gs_amd import ops
    D, cam = frame["D"], frame["cam"]
    sp2, _, gb2, ts2, sg2, st2 = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"],
                                                      D["sh"], DEG)
    assert torch.equal(ts2, frame["ts"]) and torch.equal(sg2[:frame["n"]], frame["sg"][:frame["n"]])
    rgb2, depth2, fT2, _ = ops.rasterize_fwd(cam, sp2, sg2, ts2)
    assert torch.equal(rgb2, frame["rgb"]) and torch.equal(depth2, frame["depth"]) and torch.equal(fT2, frame["fT"])
    sp3 = ops.project_fwd(cam, D["means"], D["log_scales"], D["quats"], D["opac_logit"], D["sh"], DEG)
    gb3, ts3, sg3, st3 = ops.bin_sort(cam, sp3)
    assert torch.equal(ts3, frame["ts"]) and torch.equal(sg3[:frame["n"]], frame["sg"][:frame["n"]])


def test_backward_is_linear_in_upstream_gradient(frame):
    """partials(v1 + v2) == partials(v1) + partials(v2) and the per-Gaussian reduction of the
    partials is reproducible bit for bit."""
    from touch_gs_amd import ops
    cam, sp, gb, sg, ts = frame["cam"], frame["sp"], frame["gb"], frame["sg"], frame["ts"]
    g = torch.Generator(device="cpu").manual_seed(3)
    dev = sp.device
    v = [(torch.randn(H, W, 3, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev))
         for _ in range(2)]
    def run(vr, vd, va):
        p, _ = ops.rasterize_bwd(cam, sp, gb, sg, ts, frame["rgb"], frame["depth"], frame["fT"], vr, vd, va)
        return ops.reduce_partials(cam, sp, gb, p)
    a, b = run(*v[0]), run(*v[1])
    c = run(v[0][0] + v[1][0], v[0][1] + v[1][1], v[0][2] + v[1][2])
    scale = c.abs().max(dim=0).values.clamp_min(1e-12)
    assert float(((a + b - c).abs() / scale).max()) < 2e-4
    assert torch.equal(run(*v[0]), a)


def test_zero_opacity_gaussians_change_nothing(frame):
    from touch_gs_amd import ops
    D, cam = frame["D"], frame["cam"]
    op = D["opac_logit"].clone()
    op[::3] = -30.0                       # sigmoid -> ~1e-13 < 1/255: can never contribute
    spA, _, gbA, tsA, sgA, _ = ops.project_bin_sort(cam, D["means"], D["log_scales"], D["quats"], op, D["sh"], DEG)
    rA, dA, tA, _ = ops.rasterize_fwd(cam, spA, sgA, tsA)
    keep = torch.ones(N, dtype=torch.bool, device=op.device)
    keep[::3] = False
    spB, _, gbB, tsB, sgB, _ = ops.project_bin_sort(cam, D["means"][keep].contiguous(), D["log_scales"][keep].contiguous(),
                                                    D["quats"][keep].contiguous(), op[keep].contiguous(),
                                                    D["sh"][keep].contiguous(), DEG)
    rB, dB, tB, _ = ops.rasterize_fwd(cam, spB, sgB, tsB)
    assert torch.equal(rA, rB) and torch.equal(dA, dB) and torch.equal(tA, tB)
