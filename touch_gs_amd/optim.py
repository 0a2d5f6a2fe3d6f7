"""Flat SoA parameter store + fused Adam (K9) for the Gaussian model.

All parameters of the model live in ONE flat fp32 buffer (and the gradients / Adam moments in
three more of the same layout) so that (a) the data-parallel gradient exchange is a single RCCL
collective over one buffer and (b) the optimizer is a single streaming launch.  Layout
(include/tgs.h, TgsAdamSpec):  means[3N] | log_scales[3N] | quats[4N] | opac_logit[N] | sh[N*K*3],
each segment starting at a multiple of 4 floats.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib
from ._lib import check, ptr


def _al4(x: int) -> int:
    return (x + 3) & ~3


def layout(N: int, K: int) -> Dict[str, tuple]:
    """name -> (offset, numel, shape); 'total' -> padded element count."""
    o_means = 0
    o_scales = _al4(o_means + 3 * N)
    o_quats = _al4(o_scales + 3 * N)
    o_opac = o_quats + 4 * N
    o_sh = _al4(o_opac + N)
    total = _al4(o_sh + 3 * K * N)
    return dict(means=(o_means, 3 * N, (N, 3)), log_scales=(o_scales, 3 * N, (N, 3)),
                quats=(o_quats, 4 * N, (N, 4)), opac_logit=(o_opac, N, (N,)),
                sh=(o_sh, 3 * K * N, (N, K, 3)), total=total)


class GaussianParams:
    """Flat parameter + gradient buffers with named views (SURVEY App. B.0 parameters)."""

    NAMES = ("means", "log_scales", "quats", "opac_logit", "sh")

    def __init__(self, flat: torch.Tensor, N: int, K: int):
        self.N, self.K = N, K
        self.flat = flat
        self.grad = torch.zeros_like(flat)
        for k, v in self.views_of(self.flat, N, K).items():
            setattr(self, k, v)
        self.g = self.views_of(self.grad, N, K)

    @staticmethod
    def views_of(flat: torch.Tensor, N: int, K: int) -> Dict[str, torch.Tensor]:
        L = layout(N, K)
        return {k: flat[L[k][0]:L[k][0] + L[k][1]].view(L[k][2]) for k in GaussianParams.NAMES}

    @staticmethod
    def allocate(N: int, K: int, device) -> "GaussianParams":
        flat = torch.zeros(layout(N, K)["total"], dtype=torch.float32, device=device)
        return GaussianParams(flat, N, K)

    @staticmethod
    def from_tensors(means, log_scales, quats, opac_logit, sh) -> "GaussianParams":
        N, K = means.shape[0], sh.shape[1]
        gp = GaussianParams.allocate(N, K, means.device)
        for k, t in zip(GaussianParams.NAMES, (means, log_scales, quats, opac_logit, sh)):
            getattr(gp, k).copy_(t)
        return gp

    def tensors(self):
        return tuple(getattr(self, k) for k in self.NAMES)

    @torch.no_grad()
    def permute_(self, perm: torch.Tensor, *others: torch.Tensor) -> None:
        """Reorder the Gaussians in place: row i becomes old row perm[i], in the parameter and gradient
        buffers and in every flat buffer of the same layout passed in ``others`` (Adam moments)."""
        for flat in (self.flat, self.grad) + tuple(others):
            v = GaussianParams.views_of(flat, self.N, self.K)
            for k in self.NAMES:
                v[k].copy_(v[k][perm])


    def grad_views(self):
        """Pre-allocated gradient outputs for ops.project_bwd(out=...)."""
        return tuple(self.g[k] for k in self.NAMES)


def morton_order(means: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation that sorts points along a 3-D Morton (Z-order) curve over their bounding box:
    consecutive Gaussians are spatial neighbours from every viewpoint."""
    m = means.detach().double()
    lo, hi = m.min(0).values, m.max(0).values
    q = ((m - lo) / (hi - lo).clamp_min(1e-12) * (2 ** bits - 1)).long().clamp_(0, 2 ** bits - 1)

    def spread(x):   # 10 bits -> every third bit
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(code, stable=True)


def balanced_order(means: torch.Tensor, hits: "torch.Tensor | None", group: int = 256, long_run: int = 32) -> torch.Tensor:
    """Morton order with the LONG-RUN Gaussians dealt evenly over the binning groups (round 6).

    A binning group is ``group`` consecutive rows, and three kernels spend time proportional to the (tile, Gaussian)
    pairs of their group: K1's tile counting, ``k_fill_bins`` and K8's segmented sum of the partial gradients.  In a
    converged object-centric model (the reference's kind of scene: an object on a table, 1280 x 720) 7 % of the
    Gaussians -- table, background -- cover more than 32 tiles each and hold 60 % of all pairs; plain Morton order puts
    spatial neighbours, i.e. those, into the SAME groups: one group of 28 000 pairs next to a mean of 2 600, and every
    one of the three launches lasts as long as that group (profiles/r6_before_*: K8 370 - 440 us, fill 85 us for 79 -
    145 k Gaussians against 120 / 26 us for cfg3's 1 M).  Here the Gaussians whose last frame covered more than
    ``long_run`` tiles (``hits``: tiles per Gaussian in any recent view; None or nothing long = plain Morton order) are
    sorted by that count and dealt round robin to the groups -- group g takes ranks g, g + G, g + 2 G, ...: one from
    every size stratum -- at the END of each group, behind its Morton-ordered short-run rows.  The kernels keep the
    long runs out of a group's counting box (tgs_binning.h), so the short-run rows keep their aggregated counting.  A
    pure re-layout like ``morton_order``: every kernel computes the same thing on permuted rows.  Returns the
    permutation (new row i = old row perm[i])."""
    N = means.shape[0]
    mort = morton_order(means)
    if hits is None or hits.numel() != N or N < 2 * group:
        return mort
    hits = hits.to(means.device).long()
    G = N // group                                  # full groups (a partial last group takes only short-run rows)
    big = hits > long_run
    n_big = int(big.sum())
    if n_big == 0:
        return mort
    if n_big > G * (group // 2):                    # keep at least half of every group for the Morton-ordered rows
        thr = torch.sort(hits, descending=True).values[G * (group // 2)]
        big = hits > max(int(thr), long_run)
        n_big = int(big.sum())
        if n_big == 0:
            return mort
    inv = torch.empty_like(mort)
    inv[mort] = torch.arange(N, device=mort.device)
    # size ranks, ties by Morton position (deterministic)
    bi = torch.nonzero(big).squeeze(1)
    key = (-hits[bi]) * (N + 1) + inv[bi]
    bi = bi[torch.argsort(key)]
    r = torch.arange(n_big, device=mort.device)
    g_of, k_of = r % G, r // G                      # group and position among the group's long-run rows
    c = torch.bincount(g_of, minlength=G)           # long-run rows per full group
    sizes = torch.full((G + (1 if N % group else 0),), group, dtype=torch.long, device=mort.device)
    if N % group:
        sizes[-1] = N % group
        c = torch.cat([c, c.new_zeros(1)])
    short_slots = sizes - c                         # Morton rows per group, at the front of the group
    starts = torch.cumsum(sizes, 0) - sizes
    perm = torch.empty(N, dtype=torch.long, device=mort.device)
    perm[starts[g_of] + short_slots[g_of] + k_of] = bi
    small = mort[~big[mort]]                        # short-run rows in Morton order
    sg = torch.repeat_interleave(torch.arange(len(sizes), device=mort.device), short_slots)
    first = torch.cumsum(short_slots, 0) - short_slots
    perm[starts[sg] + (torch.arange(len(small), device=mort.device) - first[sg])] = small
    return perm


class FusedAdam:
    """torch.optim.Adam semantics (no weight decay, no amsgrad) in one HIP launch (tgs_adam_step).

    ``lrs`` keys: means, log_scales, quats, opac_logit, sh_dc, sh_rest (Splatfacto's groups,
    SURVEY App. A.3).  ``grad_scale`` multiplies the gradient on the fly (e.g. 1/world_size after a
    sum all-reduce).
    """

    def __init__(self, params: GaussianParams, lrs: Dict[str, float], betas=(0.9, 0.999), eps=1e-15):
        self.p = params
        self.lrs = dict(lrs)
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        self.t = 0
        # device copy of the step's bias corrections: read by the launches recorded while
        # ``use_device_bias_corr`` is set (a captured step graph), refreshed before every replay
        self._dyn = None
        self.use_device_bias_corr = False

    def _spec(self):
        s = _lib.TgsAdamSpec()
        s.lr_means, s.lr_scales, s.lr_quats = self.lrs["means"], self.lrs["log_scales"], self.lrs["quats"]
        s.lr_opac, s.lr_sh_dc, s.lr_sh_rest = self.lrs["opac_logit"], self.lrs["sh_dc"], self.lrs["sh_rest"]
        s.beta1, s.beta2, s.eps = self.betas[0], self.betas[1], self.eps
        s.bias_corr1 = 1.0 - self.betas[0] ** self.t
        s.bias_corr2 = 1.0 - self.betas[1] ** self.t
        s.device_bias_corr = self._dyn.data_ptr() if (self.use_device_bias_corr and self._dyn is not None) else None
        return s

    def upload_bias_corr(self):
        """Write the current step's {1-beta1^t, 1-beta2^t, lr_means} to the device scalars (stream
        ordered, no host sync): called once per step before a captured step graph is replayed."""
        if self._dyn is None:
            self._dyn = torch.ones(4, dtype=torch.float32, device=self.p.flat.device)
        vals = (C.c_float * 3)(1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t, self.lrs["means"])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(_lib.load().tgs_store_small(ptr(self._dyn), vals, 3, stream), "tgs_store_small")

    def begin_step(self):
        """Advance the step counter once per optimizer iteration (before step_range calls)."""
        self.t += 1

    def step_range(self, elem_begin: int, elem_end: int, grad_scale: float = 1.0, guard=None):
        """Adam on flat elements [elem_begin, elem_end) (multiples of 4) of the current step.
        ``guard``: the frame's binning status word; the launch is a no-op if its overflow flag is set."""
        lib = _lib.load()
        s = self._spec()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.tgs_adam_step(self.p.N, self.p.K, ptr(self.p.flat), ptr(self.p.grad), ptr(self.exp_avg),
                                ptr(self.exp_avg_sq), C.byref(s), C.c_float(grad_scale), elem_begin, elem_end,
                                ptr(guard), stream), "tgs_adam_step")

    def step_geom_and_project_next(self, sh_deg: int, grad_scale: float, guard, prefetch):
        """Data-parallel step: Adam on the geometry segments from ``params.grad`` (all-reduced) fused with the NEXT
        view's K1 into ``prefetch.front`` (tgs_adam_geom_project_next); call after the SH rows have been stepped."""
        lib = _lib.load()
        s = self._spec()
        ncs, fb, budget = prefetch.cam.c_struct(), prefetch.front, prefetch.front_budget
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.tgs_adam_geom_project_next(
            C.byref(ncs), self.p.N, self.p.K, sh_deg, ptr(self.p.flat), ptr(self.p.grad), ptr(self.exp_avg),
            ptr(self.exp_avg_sq), C.byref(s), C.c_float(grad_scale), ptr(guard), ptr(prefetch.tag_word), prefetch.tag,
            ptr(fb.splats), ptr(fb.radii), ptr(fb.group_base), ptr(fb.tile_cursor), fb.cap, ptr(fb.scratch),
            ptr(fb.status), ptr(budget.sticky_word(self.p.flat.device)), int(fb.cleared), stream),
              "tgs_adam_geom_project_next")
        prefetch.front_issued = True

    def step_sh_gathered_geom_and_project_next(self, world: int, sh_deg: int, rows, blocks_all, grad_scale: float, guard,
                                               prefetch):
        """The whole tail of a data-parallel step in ONE launch (tgs_adam_sh_gathered_geom_project_next): Adam on the SH
        rows of every chunk ``rows[c] = (begin, end)`` from its all-gathered colour blocks ``blocks_all[c]`` [world, 3 rows
        + 4], Adam on the geometry segments from ``params.grad`` (all-reduced), the next view's colours from the updated
        rows while they are on chip, and its K1 into ``prefetch.front``.  Bit-identical to ``step_sh_gathered`` per chunk
        followed by ``step_geom_and_project_next``."""
        lib = _lib.load()
        s = self._spec()
        n = len(rows)
        begins = (C.c_int32 * (n + 1))(*([r[0] for r in rows] + [rows[-1][1]]))
        blks = (C.c_void_p * n)(*[b.data_ptr() for b in blocks_all])
        ncs, fb, budget = prefetch.cam.c_struct(), prefetch.front, prefetch.front_budget
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.tgs_adam_sh_gathered_geom_project_next(
            C.byref(ncs), world, self.p.N, self.p.K, sh_deg, ptr(self.p.flat), ptr(self.p.grad), n, begins, blks,
            ptr(self.exp_avg), ptr(self.exp_avg_sq), C.byref(s), C.c_float(grad_scale), ptr(guard), ptr(prefetch.tag_word),
            prefetch.tag, ptr(fb.splats), ptr(fb.radii), ptr(fb.group_base), ptr(fb.tile_cursor), fb.cap, ptr(fb.scratch),
            ptr(fb.status), ptr(budget.sticky_word(self.p.flat.device)), int(fb.cleared), stream),
              "tgs_adam_sh_gathered_geom_project_next")
        prefetch.front_issued = True

    def step(self, grad_scale: float = 1.0, guard=None):
        self.begin_step()
        self.step_range(0, -1, grad_scale, guard)

    def can_fuse_with_backward(self, sh_deg: int) -> bool:
        """tgs_project_bwd_adam needs 4 or 16 stored SH bases per Gaussian (storage degree 1 or 3);
        the active degree may be anything the storage holds (rows above it get a zero gradient)."""
        return self.p.K in (4, 16) and 0 <= sh_deg and (sh_deg + 1) ** 2 <= self.p.K

    def backward_and_step(self, cam, sh_deg: int, splats, group_base, partials, want_v_xy: bool = False,
                          begin: bool = True, guard=None, prefetch=None):
        """K8 + K9 in one launch (single-process training): gradients never reach HBM.
        ``prefetch`` = ops.ColorPrefetch armed for the next view: the kernel also writes the colours
        the UPDATED Gaussians show to that camera (tgs_project_bwd_adam_next)."""
        lib = _lib.load()
        if begin:
            self.begin_step()
        s = self._spec()
        cs = cam.c_struct()
        v_xy = torch.empty(self.p.N, 2, dtype=torch.float32, device=self.p.flat.device) if want_v_xy else None
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if prefetch is None:
            check(lib.tgs_project_bwd_adam(C.byref(cs), self.p.N, self.p.K, sh_deg, ptr(self.p.flat), ptr(self.exp_avg),
                                           ptr(self.exp_avg_sq), C.byref(s), ptr(splats), ptr(group_base),
                                           ptr(partials), ptr(v_xy), ptr(guard), stream), "tgs_project_bwd_adam")
        elif prefetch.front is not None:
            # ... and the next view's K1 (records, pair ranges, tile counts) into that frame's buffers
            ncs, fb, budget = prefetch.cam.c_struct(), prefetch.front, prefetch.front_budget
            check(lib.tgs_project_bwd_adam_next_front(
                C.byref(cs), self.p.N, self.p.K, sh_deg, ptr(self.p.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                C.byref(s), ptr(splats), ptr(group_base), ptr(partials), ptr(v_xy), ptr(guard), C.byref(ncs),
                ptr(prefetch.colors), ptr(prefetch.tag_word), prefetch.tag, ptr(fb.splats), ptr(fb.radii),
                ptr(fb.group_base), ptr(fb.tile_cursor), fb.cap, ptr(fb.scratch), ptr(fb.status),
                ptr(budget.sticky_word(self.p.flat.device)), int(fb.cleared), stream), "tgs_project_bwd_adam_next_front")
            prefetch.front_issued = True
        else:
            ncs = prefetch.cam.c_struct()
            check(lib.tgs_project_bwd_adam_next(C.byref(cs), self.p.N, self.p.K, sh_deg, ptr(self.p.flat),
                                                ptr(self.exp_avg), ptr(self.exp_avg_sq), C.byref(s), ptr(splats),
                                                ptr(group_base), ptr(partials), ptr(v_xy), ptr(guard),
                                                C.byref(ncs), ptr(prefetch.colors), ptr(prefetch.tag_word),
                                                prefetch.tag, stream), "tgs_project_bwd_adam_next")
        return v_xy

    def can_gather_sh(self) -> bool:
        """tgs_adam_step_sh_gathered streams SH rows as float4s: 3K must be a multiple of 4."""
        return (3 * self.p.K) % 4 == 0

    def geom_end(self) -> int:
        """First flat element of the SH segment (= number of geometry elements incl. padding)."""
        return layout(self.p.N, self.p.K)["sh"][0]

    def step_sh_gathered(self, world: int, sh_deg: int, v_color_all: torch.Tensor, grad_scale: float, guard=None,
                         rows=None):
        """Adam on the SH segment from the all-gathered colour-gradient blocks [world, 3N+4] of the
        current step (call before the geometry segments are stepped: it reads the means).
        ``guard``: the status word written by ops.dp_agree_overflow (no-op if any rank overflowed).
        ``rows`` = (begin, end): only these model rows, ``v_color_all`` = [world, 3 (end - begin) + 4]."""
        lib = _lib.load()
        b, e = (0, self.p.N) if rows is None else rows
        if v_color_all.numel() != world * (3 * (e - b) + 4):
            raise ValueError("v_color_all must hold world blocks of 3*rows+4 floats")
        s = self._spec()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.tgs_adam_step_sh_gathered_rows(world, self.p.N, b, e, self.p.K, sh_deg, ptr(self.p.flat),
                                                 ptr(v_color_all), ptr(self.exp_avg), ptr(self.exp_avg_sq), C.byref(s),
                                                 C.c_float(grad_scale), ptr(guard), stream),
              "tgs_adam_step_sh_gathered_rows")

    def state_dict(self):
        return dict(t=self.t, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lrs=self.lrs)

    def load_state_dict(self, sd):
        self.t = sd["t"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lrs = dict(sd["lrs"])
