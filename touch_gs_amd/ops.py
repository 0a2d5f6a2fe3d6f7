"""Operator layer: thin wrappers over the C ABI (include/tgs.h) + the autograd Functions.

Mirrors the operator surface the reference's training loop reaches through
``ns-train depth-gaussian-splatting`` (reference ``scripts/train_bunny_real.sh:52``):

* fused fast path  -- :func:`render` -> (rgb, depth_acc, alpha) in ONE compositing pass;
* gsplat-0.1 shaped -- :func:`project_gaussians`, :func:`rasterize_gaussians`,
  :func:`spherical_harmonics` (these are what nerfstudio's Splatfacto calls);
* INRIA shaped      -- ``touch_gs_amd.rasterizer.GaussianRasterizer`` (App. A.1).

Every op runs the HIP kernels; there is no eager/CPU fallback (tensors must be on the device).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import PARTIAL_FLOATS, SPLAT_FLOATS, check, ptr
from .camera import Camera


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# TGS_SLOT_OK=1: K6 leaves per-quadrant bitmaps of the Gaussians that changed a quadrant's state and K7
# skips the rest (bit-identical results).  OFF by default -- measured at cfg3, same box: K7 -3.3 % (-14 us)
# but K6 +5 % (+9.5 us: four scalar instructions per quadrant evaluation in a kernel that is issue-bound),
# i.e. 0.4 % of the step and a slower render-only path.
import os as _os
SLOT_OK = _os.environ.get("TGS_SLOT_OK", "0") == "1"


def _f32c(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# capacity management for the (tile, Gaussian) intersection buffers
# ------------------------------------------------------------------------------------------------
class IntersectBudget:
    """Remembers how many (tile, Gaussian) intersections recent frames needed.

    ``sync=True`` (default): after binning, the 8-byte status word is read back (one D2H sync per
    frame) and the frame is re-binned with a larger buffer if it overflowed -- always correct.
    ``sync=False``: no read-back; the caller must size ``capacity`` itself and call
    :meth:`check` at a convenient sync point (bench.py does this once after the timed region).
    ``speculative=True`` (implies ``sync=False``): additionally hands the kernels a persistent
    "sticky overflow" word: after an overflowing frame every later frame is empty and the guarded
    optimizer kernels are no-ops, so the caller may read the status words late, grow the buffers,
    clear the word and replay (``DepthGaussianSplattingModel.enable_speculative_budget``).
    """

    def __init__(self, capacity: int = 0, sync: bool = True, growth: float = 1.25, speculative: bool = False,
                 max_list_hint: int = -1):
        self.capacity = int(capacity)
        # the caller's bound on the longest tile list of the frames binned with this budget (tgs.h, max_list_hint):
        # -1 = none, every sort class is launched; a frame that breaks it is void like one that overflowed (sticky
        # word, replay) -- so only budgets that do NOT read the status back per frame may carry one
        self.max_list_hint = int(max_list_hint)
        self.sync = sync and not speculative
        self.speculative = speculative
        self.growth = growth
        self.last_status = None  # device int32[2] of the most recent frame: {#intersections, overflow}
        self.last_status4 = None  # ... and the full word {.., .., required capacity, reserved}
        self.last_n = None       # #intersections of the last frame that was read back
        self.last_need = None    # capacity that frame needs under the per-XCD split (>= last_n; tgs.h)
        self.last_longest = None  # longest tile list of that frame (status[3])
        self.sticky = None       # device int32[1], allocated on first use

    def sticky_word(self, device):
        """Persistent overflow word of every budget that does not read the status back (``sync=False``):
        an overflowing frame sets it, later frames start overflowed (empty lists) and the guarded
        optimizer kernels do nothing -- an undersized buffer can never feed garbage partials to Adam,
        and :meth:`check` sees an overflow of ANY earlier frame, not only of the last one."""
        if self.sync:
            return None
        if self.sticky is None:
            if torch.cuda.is_current_stream_capturing():
                # allocated inside a capture the word would live in the graph's pool and its zero-fill
                # would be a graph node: every replay would forget an earlier overflow
                raise RuntimeError("allocate the sticky overflow word (budget.sticky_word(device)) before graph capture")
            self.sticky = torch.zeros(1, dtype=torch.int32, device=device)
        return self.sticky

    def list_hint(self) -> int:
        return -1 if self.sync else self.max_list_hint

    def initial(self, N: int):
        if self.capacity <= 0:
            self.capacity = max(8 * N, 1 << 16)
        return self.capacity

    def check(self):
        """Read the last frame's status (and the sticky word, which remembers an overflow of any
        earlier frame of a ``sync=False`` budget); raises on overflow.  Returns #intersections."""
        if self.last_status is None:
            return None
        if self.last_status4 is not None:
            n, ovf, need, longest = self.last_status4.tolist()
            self.last_need = max(n, need)
            self.last_longest = longest
            if ovf and 0 <= self.max_list_hint < longest:
                raise RuntimeError(f"a tile list of {longest} entries broke max_list_hint = {self.max_list_hint}: that frame "
                                   "and every frame since were dropped (raise the hint or pass -1)")
        else:
            n, ovf = self.last_status.tolist()
        self.last_n = n
        if self.sticky is not None and int(self.sticky.item()) != 0:
            ovf = 1
        if ovf:
            # the sticky word remembers ANY earlier frame: with a list hint in force the cause may have been a list longer
            # than the hint in a frame whose status words are gone, not the capacity
            why = (f"intersection capacity {self.capacity} too small" if self.max_list_hint < 0 else
                   f"intersection capacity {self.capacity} too small, or a tile list broke max_list_hint = {self.max_list_hint} in an earlier frame")
            raise RuntimeError(f"{why}: a frame needed more (last frame: {n} intersections, longest list "
                               f"{getattr(self, 'last_longest', '?')}); every frame since was dropped")
        return n


_default_budget = IntersectBudget()


# ------------------------------------------------------------------------------------------------
# thin wrappers (one per C entry point)
# ------------------------------------------------------------------------------------------------
def project_fwd(cam: Camera, means, log_scales, quats, opac_logit, sh, sh_deg: int, colors=None,
                want_radii: bool = False):
    """K1 -> splats [N,12] (, radii int32 [N]).  (tgs_project_fwd)"""
    lib = _lib.load()
    N = means.shape[0]
    splats = torch.empty(max(N, 1), SPLAT_FLOATS, dtype=torch.float32, device=means.device)[:N]
    radii = torch.empty(N, dtype=torch.int32, device=means.device) if want_radii else None
    cs = cam.c_struct()
    sh_stride = sh.shape[1] if sh is not None else 0
    check(lib.tgs_project_fwd(C.byref(cs), N, ptr(means), ptr(log_scales), ptr(quats), ptr(opac_logit),
                              ptr(sh), sh_stride, sh_deg if sh is not None else -1, ptr(colors),
                              ptr(splats), ptr(radii), _stream()), "tgs_project_fwd")
    return (splats, radii) if want_radii else splats


def bin_sort(cam: Camera, splats, budget: Optional[IntersectBudget] = None):
    """K2-K5 -> (group_base, tile_start, sorted_gid, status).  (tgs_bin_sort)"""
    lib = _lib.load()
    budget = budget or _default_budget
    N = splats.shape[0]
    dev = splats.device
    T = cam.num_tiles
    G = lib.tgs_num_groups(N)
    cs = cam.c_struct()
    group_base = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    tile_start = torch.empty(T + 1 + 512, dtype=torch.int32, device=dev)[:T + 1]   # + 512 scratch ints of the rasterizer (tgs.h)
    nc = lib.tgs_tile_counter_len(cam.W, cam.H)
    counters = torch.empty(nc + 4, dtype=torch.int32, device=dev)   # per-XCD tile counters + sub-list starts | status[4]
    tile_cursor, status, status4 = counters[:nc], counters[nc:nc + 2], counters[nc:]
    # block -> tile schedule of K6 / K7 (tiles dealt to the XCDs in granules of 8, longest list first inside each XCD); it rides on the
    # tile_start tensor object so that the (tile_start, sorted_gid) pair keeps its meaning for callers
    tile_order = torch.empty(lib.tgs_tile_order_len(cam.W, cam.H), dtype=torch.int32, device=dev)
    tile_start.tile_order = tile_order
    tile_start.slot_ok = None          # quadrant bitmaps K6 leaves for K7 (rasterize_fwd allocates and fills them)
    cap = budget.initial(N)
    while True:
        sorted_gid = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        scratch = torch.empty(lib.tgs_sort_scratch_bytes(cap), dtype=torch.uint8, device=dev)
        check(lib.tgs_bin_sort(C.byref(cs), N, ptr(splats), ptr(group_base), ptr(tile_start), _tile_start_len(tile_start),
                               ptr(tile_cursor), ptr(sorted_gid), ptr(tile_order), cap, ptr(scratch),
                               ptr(status), ptr(budget.sticky_word(dev)), budget.list_hint(), _stream()), "tgs_bin_sort")
        budget.last_status = status
        budget.last_status4 = status4
        if not budget.sync:
            break
        n, ovf, need, budget.last_longest = status4.tolist()
        budget.last_n, budget.last_need = n, max(n, need)
        if not ovf:
            break
        cap = int(n * budget.growth) + 1024
        budget.capacity = cap
    return group_base, tile_start, sorted_gid, status


class ColorPrefetch:
    """Colours of the Gaussians for ONE upcoming view, evaluated by the previous step's fused optimizer
    kernel (tgs_project_bwd_adam_next) so that K1 does not re-read the 12K-byte SH rows
    (tgs_project_bin_sort_colors).  ``tag_word`` (device int32) holds ``tag`` once the kernel that was
    asked for this prefetch has run to its end; K1 falls back to the SH rows otherwise."""

    def __init__(self, N: int, device):
        self.colors = torch.empty(max(N, 1), 3, dtype=torch.float32, device=device)
        self.tag_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.tag = 0
        self.cam = None
        self.N, self.sh_deg = N, -1
        self.front = None           # FrontBuffers of that view's frame, if its K1 is prefetched too
        self.front_budget = None    # the IntersectBudget that frame is binned with (capacity, sticky word)
        self.front_issued = False   # the fused optimizer call that fills `front` has been enqueued
        self.colors_valid = True    # False: only `front` is produced (data-parallel form), `colors` stays unwritten

    def arm(self, cam: Camera, sh_deg: int, front: Optional["FrontBuffers"] = None,
            budget: Optional["IntersectBudget"] = None, colors_valid: bool = True) -> "ColorPrefetch":
        self.tag += 1
        self.cam, self.sh_deg = cam, sh_deg
        self.front, self.front_budget, self.front_issued = front, budget, False
        self.colors_valid = colors_valid
        return self

    def matches(self, cam: Camera, N: int, sh_deg: int) -> bool:
        return self.cam is cam and self.N == N and self.sh_deg == sh_deg and self.tag > 0


_SIDE_CLEAR = _os.environ.get("TGS_FRONT_SIDE_CLEAR", "1") != "0"   # A/B switch: next frame's counters cleared by extra workgroups of this frame's scan launch


class FrontBuffers:
    """Everything the front half of ONE frame writes (what project_bin_sort allocates per call).  Allocated one
    step ahead when the previous step's fused optimizer kernel also runs this frame's K1 (front prefetch,
    tgs_project_bwd_adam_next_front): records, group bases, per-tile counters and ranks are then filled by that
    kernel, and project_bin_sort only scans, fills and sorts (tgs_project_bin_sort_front)."""

    def __init__(self, cam: Camera, N: int, cap: int, want_radii: bool, dev):
        lib = _lib.load()
        T = cam.num_tiles
        self.N, self.cap, self.cam = N, cap, cam
        self.splats = torch.empty(max(N, 1), SPLAT_FLOATS, dtype=torch.float32, device=dev)[:N]  # non-null even for N = 0
        self.radii = torch.empty(N, dtype=torch.int32, device=dev) if want_radii else None
        self.group_base = torch.empty(max(lib.tgs_num_groups(N), 1), dtype=torch.int32, device=dev)
        self.tile_start = torch.empty(T + 1 + 512, dtype=torch.int32, device=dev)[:T + 1]   # + 512 scratch ints of the rasterizer (tgs.h)
        nc = lib.tgs_tile_counter_len(cam.W, cam.H)
        counters = torch.empty(nc + 4, dtype=torch.int32, device=dev)   # per-XCD tile counters + sub-list starts | status[4]
        self.tile_cursor, self.status, self.status4 = counters[:nc], counters[nc:nc + 2], counters[nc:]
        # block -> tile schedule of K6 / K7 (tiles dealt to the XCDs in granules of 8, longest list first inside each XCD); it rides on the
        # tile_start tensor object so that the (tile_start, sorted_gid) pair keeps its meaning for callers
        self.tile_start.tile_order = torch.empty(lib.tgs_tile_order_len(cam.W, cam.H), dtype=torch.int32, device=dev)
        self.tile_start.slot_ok = None     # quadrant bitmaps K6 leaves for K7 (rasterize_fwd allocates and fills them)
        self.sorted_gid = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        self.scratch = torch.empty(lib.tgs_sort_scratch_bytes(cap), dtype=torch.uint8, device=dev)
        self.cleared = False     # counters + status word already cleared (by the previous frame's scan launch)


def project_bin_sort(cam: Camera, means, log_scales, quats, opac_logit, sh, sh_deg: int,
                     budget: Optional[IntersectBudget] = None, want_radii: bool = False,
                     colors: Optional[ColorPrefetch] = None, next_front: Optional["FrontBuffers"] = None):
    """K1 fused with the tile counting, then scan / fill / sort: the front half of a frame in ONE C
    call -> (splats, radii or None, group_base, tile_start, sorted_gid, status).  (tgs_project_bin_sort;
    with ``colors`` -- a ColorPrefetch armed for THIS camera -- tgs_project_bin_sort_colors, or, if the previous
    step's optimizer kernel also ran this frame's K1 into ``colors.front``, tgs_project_bin_sort_front: scan, fill
    and sort only.  ``next_front``: the FrontBuffers of the frame AFTER this one, if this step's optimizer kernel is
    going to fill them -- the front path clears their counters on the side)"""
    lib = _lib.load()
    budget = budget or _default_budget
    N = means.shape[0]
    dev = means.device
    cs = cam.c_struct()
    sh_stride = sh.shape[1] if sh is not None else 0
    cap = budget.initial(N)
    # front prefetch: this frame's K1 already ran inside the previous step's optimizer kernel, into buffers
    # allocated then (same sizes as here) -- usable if nothing changed in between
    fb = colors.front if (colors is not None and sh is not None and N > 0 and colors.front_issued) else None
    if fb is not None:
        colors.front, colors.front_issued = None, False      # single use: the finish below consumes the counters
        if (fb.N != N or fb.cap != cap or fb.cam is not cam or colors.front_budget is not budget
                or (want_radii and fb.radii is None)):
            fb = None
    from_front = fb is not None
    tag_holder = colors
    if colors is not None and not colors.colors_valid:
        colors = None                    # nothing but the front was prefetched: any other path evaluates the SH rows
    if fb is None:
        fb = FrontBuffers(cam, N, cap, want_radii, dev)
    splats, radii, group_base, tile_start = fb.splats, fb.radii, fb.group_base, fb.tile_start
    tile_cursor, status, status4, tile_order = fb.tile_cursor, fb.status, fb.status4, fb.tile_start.tile_order
    sorted_gid, scratch = fb.sorted_gid, fb.scratch
    while True:
        if from_front:
            nf = next_front if (_SIDE_CLEAR and next_front is not None and not next_front.cleared and next_front is not fb and
                                lib.tgs_front_can_clear_next(N, next_front.cam.W, next_front.cam.H)) else None
            ncs = nf.cam.c_struct() if nf is not None else None
            check(lib.tgs_project_bin_sort_front(C.byref(cs), N, ptr(means), ptr(log_scales), ptr(quats), ptr(opac_logit),
                                                 ptr(sh), sh_stride, sh_deg, ptr(splats), ptr(radii), ptr(group_base),
                                                 ptr(tile_start), _tile_start_len(tile_start), ptr(tile_cursor), ptr(sorted_gid),
                                                 ptr(tile_order), cap, ptr(scratch), ptr(status), ptr(budget.sticky_word(dev)),
                                                 budget.list_hint(), ptr(tag_holder.tag_word), tag_holder.tag,
                                                 C.byref(ncs) if nf is not None else None,
                                                 ptr(nf.tile_cursor) if nf is not None else None,
                                                 ptr(nf.status) if nf is not None else None, _stream()),
                  "tgs_project_bin_sort_front")
            if nf is not None:
                nf.cleared = True
            from_front = False          # a regrown capacity (synchronous budget) goes through the regular K1
        elif colors is None or sh is None:
            check(lib.tgs_project_bin_sort(C.byref(cs), N, ptr(means), ptr(log_scales), ptr(quats), ptr(opac_logit),
                                           ptr(sh), sh_stride, sh_deg if sh is not None else -1, ptr(splats),
                                           ptr(radii), ptr(group_base), ptr(tile_start), _tile_start_len(tile_start),
                                           ptr(tile_cursor), ptr(sorted_gid), ptr(tile_order), cap, ptr(scratch), ptr(status),
                                           ptr(budget.sticky_word(dev)), budget.list_hint(), _stream()),
                  "tgs_project_bin_sort")
        else:
            check(lib.tgs_project_bin_sort_colors(C.byref(cs), N, ptr(means), ptr(log_scales), ptr(quats),
                                                  ptr(opac_logit), ptr(sh), sh_stride, sh_deg, ptr(splats),
                                                  ptr(radii), ptr(group_base), ptr(tile_start), _tile_start_len(tile_start),
                                                  ptr(tile_cursor), ptr(sorted_gid), ptr(tile_order), cap, ptr(scratch), ptr(status),
                                                  ptr(budget.sticky_word(dev)), budget.list_hint(), ptr(colors.colors),
                                                  ptr(colors.tag_word), colors.tag, _stream()),
                  "tgs_project_bin_sort_colors")
        budget.last_status = status
        budget.last_status4 = status4
        if not budget.sync:
            break
        n, ovf, need, budget.last_longest = status4.tolist()
        budget.last_n, budget.last_need = n, max(n, need)
        if not ovf:
            break
        cap = int(n * budget.growth) + 1024
        budget.capacity = cap
        sorted_gid = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        scratch = torch.empty(lib.tgs_sort_scratch_bytes(cap), dtype=torch.uint8, device=dev)
    return splats, radii, group_base, tile_start, sorted_gid, status


def _tile_start_len(tile_start) -> int:
    """int32 entries of the allocation from the tensor's first element on: the ``tile_start_len`` the C ABI validates
    against ``tgs_tile_start_len`` (TGS_VERSION 300)."""
    return tile_start.untyped_storage().nbytes() // 4 - tile_start.storage_offset()


def raster_opts(k6_blocks=None, k6_split=None, k7_front_to_back=None, k7_quad=None, k7_quad_min_walk=None, k7_blocks=None):
    """Per-call TgsRasterOpts (tgs.h): the forms of the compositing kernels for ONE rasterize_fwd / rasterize_bwd call,
    independent of the process-wide ``set_raster_variant`` / ``set_k6_split`` / ``set_k7_quad`` defaults.  None = default."""
    f = lambda v: -1 if v is None else int(v)
    return _lib.TgsRasterOpts(f(k6_blocks), f(k6_split), f(k7_front_to_back), f(k7_quad), f(k7_quad_min_walk), f(k7_blocks))


def _check_tile_start(tile_start, T: int) -> None:
    """The rasterizer keeps 512 scratch ints behind the T + 1 tile starts (tgs.h, tgs_tile_start_len): a tensor that was
    copied out of the buffer ``bin_sort`` / ``FrontBuffers`` allocated (clone, contiguous, .to) has lost them.  (The C
    ABI checks the same length; this gives the Python caller the reason.)"""
    have = _tile_start_len(tile_start)
    if have < T + 1 + 512:
        raise ValueError(f"tile_start must be (a view of) a buffer of T + 513 = {T + 513} int32 (got room for {have}): "
                         "pass the tensor ops.bin_sort / ops.project_bin_sort returned, not a copy")


def rasterize_fwd(cam: Camera, splats, sorted_gid, tile_start, want_idx: bool = False, want_stop: bool = True, opts=None):
    """K6 -> (rgb [H,W,3], depth_acc [H,W], final_T [H,W], final_idx [H,W] or None).  (tgs_rasterize_fwd)

    ``want_stop=False`` (render only: evaluation, ``get_outputs``) skips the per-pixel stop positions the backward needs
    (4 B per pixel not allocated, not written); ``opts`` = ``raster_opts(...)`` for this call only."""
    lib = _lib.load()
    dev = splats.device
    H, W = cam.H, cam.W
    _check_tile_start(tile_start, cam.num_tiles)
    rgb = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(H, W, dtype=torch.float32, device=dev)
    fT = torch.empty(H, W, dtype=torch.float32, device=dev)
    fidx = torch.empty(H, W, dtype=torch.int32, device=dev) if want_idx else None
    # per pixel: list position of the Gaussian that stopped it -- where the backward (back to front) starts.
    # It travels with final_T (fT.stop_pos), like tile_order / slot_ok travel with tile_start.
    stop = torch.empty(H, W, dtype=torch.int32, device=dev) if want_stop else None
    fT.stop_pos = stop
    cs = cam.c_struct()
    slot_ok = None
    if SLOT_OK and hasattr(tile_start, "slot_ok"):
        # per batch of 64 list positions: which Gaussians changed the state of each 8x8 quadrant; the
        # backward over the same lists (rasterize_bwd*) skips the rest.  Results are unaffected.
        slot_ok = torch.empty(4 * lib.tgs_slot_ok_len(W, H, sorted_gid.shape[0]), dtype=torch.int64, device=dev)
        tile_start.slot_ok = slot_ok
    check(lib.tgs_rasterize_fwd(C.byref(cs), ptr(splats), ptr(sorted_gid), ptr(tile_start), _tile_start_len(tile_start),
                                ptr(getattr(tile_start, "tile_order", None)), ptr(rgb),
                                ptr(depth), ptr(fT), ptr(fidx), ptr(stop), ptr(slot_ok),
                                C.byref(opts) if opts is not None else None, _stream()), "tgs_rasterize_fwd")
    return rgb, depth, fT, fidx


def rasterize_bwd(cam: Camera, splats, group_base, sorted_gid, tile_start, rgb, depth, fT,
                  v_rgb=None, v_depth=None, v_alpha=None, loss: Optional[dict] = None,
                  want_tile_loss: bool = False, stop_pos=None, partials=None, opts=None):
    """K7 -> (partials [cap,12], tile_loss [T,2] or None).  (tgs_rasterize_bwd)

    ``loss`` = dict(gt_rgb, gt_depth, uncertainty, l1_weight, depth_weight, uncertainty_weight, eps).
    ``stop_pos``: the forward's per-pixel stop positions; default = ``fT.stop_pos`` as ``rasterize_fwd`` left it.
    """
    lib = _lib.load()
    dev = splats.device
    _check_tile_start(tile_start, cam.num_tiles)
    if partials is None:     # (a caller may hand in the buffer, e.g. pre-filled to detect reads of unwritten records)
        partials = torch.empty(sorted_gid.shape[0], PARTIAL_FLOATS, dtype=torch.float32, device=dev)
    tile_loss = torch.empty(cam.num_tiles, 2, dtype=torch.float32, device=dev) if want_tile_loss else None
    cs = cam.c_struct()
    ls = None
    keep = []
    if loss is not None:
        ls = _loss_spec_struct(loss, keep)
    v_rgb, v_depth, v_alpha = _f32c(v_rgb), _f32c(v_depth), _f32c(v_alpha)
    check(lib.tgs_rasterize_bwd(C.byref(cs), ptr(splats), ptr(group_base), ptr(sorted_gid),
                                ptr(tile_start), _tile_start_len(tile_start), ptr(getattr(tile_start, "tile_order", None)),
                                ptr(rgb), ptr(depth), ptr(fT), ptr(_stop_pos_of(fT, stop_pos)),
                                ptr(v_rgb), ptr(v_depth), ptr(v_alpha),
                                C.byref(ls) if ls is not None else None, ptr(partials),
                                ptr(tile_loss), ptr(getattr(tile_start, "slot_ok", None)),
                                C.byref(opts) if opts is not None else None, _stream()), "tgs_rasterize_bwd")
    return partials, tile_loss


def set_raster_variant(k6_blocks: Optional[bool] = None, k7_front_to_back: Optional[bool] = None) -> int:
    """Developer switch (tgs_set_raster_variant): K6 in 4x4-block (default) / quadrant form, K7 back to front
    (default) / front to back (the round 1-3 form; the only one that takes the slot_ok bitmaps).  None leaves a
    setting alone.  Returns the settings in force (bit 0 = block-form K6, bit 1 = front-to-back K7)."""
    f = lambda v: -1 if v is None else int(bool(v))
    return _lib.load().tgs_set_raster_variant(f(k6_blocks), f(k7_front_to_back))


def set_k6_split_shape(floor: Optional[int] = None, heads: Optional[int] = None):
    """The shape of K6's split rule (tgs_set_k6_split_shape): the shortest list it splits and how many leading entries of
    the schedule get extra blocks.  None leaves a setting.  Returns (floor, heads)."""
    r = _lib.load().tgs_set_k6_split_shape(-1 if floor is None else int(floor), -1 if heads is None else int(heads))
    return r & 0xffff, r >> 16


def set_long_run(tiles: Optional[int] = None) -> int:
    """Binning: Gaussians covering more than ``tiles`` tiles are long runs (tgs_set_long_run; default 32, None = query):
    counted outside the group's aggregated box, their partial records summed by the whole workgroup in K8."""
    return _lib.load().tgs_set_long_run(-1 if tiles is None else int(tiles))


def set_k6_split(factor: Optional[int] = None) -> int:
    """The forward splits tiles whose list exceeds max(256, factor x the balanced per-slot load) into four quadrant blocks
    of the same launch (tgs_set_k6_split; default 2, 0 = never, None = query).  Bit-identical outputs."""
    return _lib.load().tgs_set_k6_split(-1 if factor is None else int(factor))


def set_k7_quad(factor: Optional[int] = None, min_walk: Optional[int] = None):
    """K7's four-waves-per-tile form for the tiles of chain-bound frames (deepest walk > factor / 2 x the balanced
    per-slot load; tgs_set_k7_quad; defaults factor 8, min_walk 16; factor 0 = one wave per tile always; None leaves
    a setting).  Returns (factor, min_walk) in effect."""
    r = _lib.load().tgs_set_k7_quad(-1 if factor is None else int(factor), -1 if min_walk is None else int(min_walk))
    return r & 255, r >> 8


def set_k7_scan(min_walk: Optional[int] = None, heads: Optional[int] = None):
    """K7's scan form for the longest tiles of chain-bound frames (tgs_set_k7_scan): tiles walking more than ``min_walk``
    entries among the schedule's first ``heads`` slots; 0 = off; None leaves a setting.  Returns (min_walk, heads)."""
    r = _lib.load().tgs_set_k7_scan(-1 if min_walk is None else int(min_walk), -1 if heads is None else int(heads))
    return r & 0xffff, r >> 16


def _stop_pos_of(fT, stop_pos=None):
    """The stop positions that belong to a forward's final_T (K7 needs them; there is no fallback)."""
    sp = stop_pos if stop_pos is not None else getattr(fT, "stop_pos", None)
    if sp is None:
        raise RuntimeError("rasterize_bwd needs the forward's stop positions: pass the final_T tensor that "
                           "rasterize_fwd returned (it carries .stop_pos) or stop_pos=...")
    return sp


def _loss_spec_struct(loss, keep):
    ls = _lib.TgsLossSpec()
    for k in ("gt_rgb", "gt_depth", "uncertainty"):
        t = _f32c(loss.get(k))
        keep.append(t)
        setattr(ls, k, ptr(t))
    ls.l1_weight = float(loss.get("l1_weight", 0.0))
    ls.depth_weight = float(loss.get("depth_weight", 0.0))
    ls.uncertainty_weight = float(loss.get("uncertainty_weight", 1.0))
    ls.eps = float(loss.get("eps", 1e-6))
    return ls


def band_rows(cam: Camera):
    """Image bands of the K7 band launches -> list of (band, y0, y1, count_y0, count_y1): the pixel rows
    [y0, y1) a band's tiles cover and the rows [count_y0, count_y1) it contributes to image-wide sums (a
    partition of [0, H)).  (tgs_num_bands / tgs_band_tiles)"""
    lib = _lib.load()
    TW = cam.tiles[0]
    out = []
    t0, t1 = C.c_int(), C.c_int()
    for b in range(lib.tgs_num_bands(cam.W, cam.H)):
        check(lib.tgs_band_tiles(cam.W, cam.H, b, C.byref(t0), C.byref(t1)), "tgs_band_tiles")
        if t1.value > t0.value:
            out.append([b, 16 * (t0.value // TW), min(16 * ((t1.value - 1) // TW + 1), cam.H)])
    for i, r in enumerate(out):
        r += [r[1] if i else 0, out[i + 1][1] if i + 1 < len(out) else cam.H]
    return [tuple(r) for r in out]


def rasterize_bwd_ssim_pipelined(cam: Camera, splats, group_base, sorted_gid, tile_start, rgb, depth, fT, gt_rgb,
                                 ssim_weight: float, loss: Optional[dict], side_stream, want_tile_loss: bool = True):
    """SSIM (K10) and K7 pipelined by image bands on two streams: the SSIM gradient of band b is produced
    on ``side_stream`` while K7 composites band b-1 on the current stream (K7 is VALU-bound, the SSIM
    kernels are latency chains: they overlap almost for free), so only the first band's SSIM is exposed.
    Bit-identical to ssim_fwd_bwd followed by rasterize_bwd.  -> (partials, tile_loss, ssim partial sums).
    (tgs_ssim_fwd_bwd_rows, tgs_rasterize_bwd_band)"""
    lib = _lib.load()
    dev = splats.device
    H, W = cam.H, cam.W
    order = getattr(tile_start, "tile_order", None)
    if order is None:
        raise ValueError("the band launches need tile_start.tile_order (ops.bin_sort / project_bin_sort)")
    bands = band_rows(cam)
    rgb, gt_rgb = _f32c(rgb), _f32c(gt_rgb)
    partials = torch.empty(sorted_gid.shape[0], PARTIAL_FLOATS, dtype=torch.float32, device=dev)
    tile_loss = torch.empty(cam.num_tiles, 2, dtype=torch.float32, device=dev) if want_tile_loss else None
    v_img = torch.empty_like(rgb)
    scratch = torch.empty(9 * H * W, dtype=torch.float32, device=dev)
    n_p = ((W + 63) // 64) * ((max(y1 - y0 for _, y0, y1, _, _ in bands) + 10) // 12 + 2)
    bp = torch.empty(len(bands), n_p, dtype=torch.float32, device=dev)
    keep = []
    ls = _loss_spec_struct(loss, keep) if loss is not None else None
    cs = cam.c_struct()
    main = torch.cuda.current_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)                     # the rendered image is complete on the current stream
    side_stream.wait_event(ready)
    for t in (v_img, scratch, bp, rgb, gt_rgb):
        t.record_stream(side_stream)
    done = []
    with torch.cuda.stream(side_stream):
        for i, (b, y0, y1, c0, c1) in enumerate(bands):
            check(lib.tgs_ssim_fwd_bwd_rows(W, H, ptr(rgb), ptr(gt_rgb), C.c_float(ssim_weight), ptr(bp[i]), n_p,
                                            ptr(v_img), ptr(scratch), y0, y1, c0, c1, _stream()),
                  "tgs_ssim_fwd_bwd_rows")
            ev = torch.cuda.Event()
            ev.record(side_stream)
            done.append(ev)
    for (b, *_), ev in zip(bands, done):
        main.wait_event(ev)
        check(lib.tgs_rasterize_bwd_band(C.byref(cs), ptr(splats), ptr(group_base), ptr(sorted_gid), ptr(tile_start),
                                         _tile_start_len(tile_start), ptr(order), ptr(rgb), ptr(depth), ptr(fT),
                                         ptr(_stop_pos_of(fT)), ptr(v_img), None, None,
                                         C.byref(ls) if ls is not None else None, ptr(partials), ptr(tile_loss),
                                         b, ptr(getattr(tile_start, "slot_ok", None)), None, _stream()), "tgs_rasterize_bwd_band")
    return partials, tile_loss, bp


def reduce_partials(cam: Camera, splats, group_base, partials):
    """K8a -> v_splats [N,12].  (tgs_reduce_partials)"""
    lib = _lib.load()
    N = splats.shape[0]
    v_splats = torch.empty(N, SPLAT_FLOATS, dtype=torch.float32, device=splats.device)
    cs = cam.c_struct()
    check(lib.tgs_reduce_partials(N, ptr(splats), ptr(group_base), C.byref(cs), ptr(partials),
                                  ptr(v_splats), _stream()), "tgs_reduce_partials")
    return v_splats


def project_bwd(cam: Camera, means, log_scales, quats, opac_logit, sh, sh_deg, splats,
                group_base=None, partials=None, v_splats=None, out=None, want_v_xy=False, guard=None):
    """K8 -> (v_means, v_log_scales, v_quats, v_opac_logit, v_sh, v_xy).  (tgs_project_bwd)

    ``out`` may supply pre-allocated gradient tensors (e.g. views into a flat gradient buffer).
    """
    lib = _lib.load()
    N = means.shape[0]
    dev = means.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    if out is None:
        out = (e(N, 3), e(N, 3), e(N, 4), e(N), torch.empty_like(sh) if sh is not None else None)
    v_means, v_ls, v_q, v_ol, v_sh = out
    v_xy = e(N, 2) if want_v_xy else None
    cs = cam.c_struct()
    sh_stride = sh.shape[1] if sh is not None else 0
    check(lib.tgs_project_bwd(C.byref(cs), N, ptr(means), ptr(log_scales), ptr(quats), ptr(opac_logit),
                              ptr(sh), sh_stride, sh_deg if sh is not None else -1, ptr(splats),
                              ptr(group_base), ptr(partials), ptr(v_splats), ptr(v_means), ptr(v_ls),
                              ptr(v_q), ptr(v_ol), ptr(v_sh), ptr(v_xy), ptr(guard), _stream()), "tgs_project_bwd")
    return v_means, v_ls, v_q, v_ol, v_sh, v_xy


def project_bwd_color(cam: Camera, means, log_scales, quats, opac_logit, sh, sh_deg, splats, group_base,
                      partials, out, v_color, want_v_xy=False, guard=None, rows=None, v_xy=None):
    """K8 of the data-parallel step (tgs_project_bwd_color): geometry gradients into ``out`` =
    (v_means, v_log_scales, v_quats, v_opac_logit) and, instead of the SH gradient, the block
    ``v_color`` [3N+4] = clamp-gated colour gradients | camera position | pad.  -> v_xy or None.
    ``rows`` = (begin, end): only these model rows (begin a multiple of 256); ``v_color`` is then the
    chunk's own block [3 (end - begin) + 4] and ``v_xy`` [N,2] (if wanted) must be passed in."""
    lib = _lib.load()
    N = means.shape[0]
    b, e = (0, N) if rows is None else rows
    if v_color.numel() != 3 * (e - b) + 4 or v_color.dtype != torch.float32:
        raise ValueError("v_color must be a float32 tensor of 3*rows+4 elements")
    v_means, v_ls, v_q, v_ol = out
    if v_xy is None and want_v_xy:
        if rows is not None:
            raise ValueError("a row-range call writes into the caller's v_xy [N,2]")
        v_xy = torch.empty(N, 2, dtype=torch.float32, device=means.device)
    cs = cam.c_struct()
    check(lib.tgs_project_bwd_color_rows(C.byref(cs), N, b, e, ptr(means), ptr(log_scales), ptr(quats), ptr(opac_logit),
                                         ptr(sh), sh.shape[1], sh_deg, ptr(splats), ptr(group_base), ptr(partials),
                                         ptr(v_means), ptr(v_ls), ptr(v_q), ptr(v_ol), ptr(v_color), ptr(v_xy),
                                         ptr(guard), _stream()), "tgs_project_bwd_color_rows")
    return v_xy


def dp_agree_overflow(world: int, N: int, v_color_all, status_out, sticky=None):
    """After the all-gather of the colour blocks: status_out = {0, any rank's frame overflowed}
    (identical on every rank); raises ``sticky`` too.  (tgs_dp_agree_overflow)"""
    check(_lib.load().tgs_dp_agree_overflow(world, N, ptr(v_color_all), ptr(status_out), ptr(sticky), _stream()),
          "tgs_dp_agree_overflow")
    return status_out


# ------------------------------------------------------------------------------------------------
# fused differentiable render
# ------------------------------------------------------------------------------------------------
class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, log_scales, quats, opac_logit, sh, means2d, cam, sh_deg, budget, need_bwd=True):
        means, log_scales, quats, opac_logit, sh = map(_f32c, (means, log_scales, quats, opac_logit, sh))
        splats, radii, group_base, tile_start, sorted_gid, _ = project_bin_sort(
            cam, means, log_scales, quats, opac_logit, sh, sh_deg, budget, want_radii=True)
        # need_bwd = False: render only (evaluation, get_outputs under no_grad; decided by render(): grad mode is always
        # off in here and needs_input_grad ignores it) -- no backward will ask for the stop positions
        rgb, depth, fT, fidx = rasterize_fwd(cam, splats, sorted_gid, tile_start, want_stop=need_bwd)
        ctx.cam, ctx.sh_deg = cam, sh_deg
        ctx.want_xy = means2d is not None
        if need_bwd:
            ctx.save_for_backward(means, log_scales, quats, opac_logit, sh, splats, group_base,
                                  tile_start, sorted_gid, rgb, depth, fT, fT.stop_pos)
        alpha = 1.0 - fT
        ctx.mark_non_differentiable(radii)
        return rgb, depth, alpha, radii

    @staticmethod
    def backward(ctx, v_rgb, v_depth, v_alpha, _v_radii):
        (means, log_scales, quats, opac_logit, sh, splats, group_base, tile_start, sorted_gid,
         rgb, depth, fT, stop) = ctx.saved_tensors
        cam = ctx.cam
        partials, _ = rasterize_bwd(cam, splats, group_base, sorted_gid, tile_start, rgb, depth, fT,
                                    v_rgb, v_depth, v_alpha, stop_pos=stop)
        v_means, v_ls, v_q, v_ol, v_sh, v_xy = project_bwd(
            cam, means, log_scales, quats, opac_logit, sh, ctx.sh_deg, splats, group_base, partials,
            want_v_xy=ctx.want_xy)
        return v_means, v_ls, v_q, v_ol, v_sh, v_xy, None, None, None, None


def render(means, log_scales, quats, opac_logit, sh, cam: Camera, sh_deg: int,
           means2d: Optional[torch.Tensor] = None, budget: Optional[IntersectBudget] = None):
    """Fused differentiable render of RGB + depth + alpha in one compositing pass.

    Parameters are the raw (pre-activation) Gaussian parameters of SURVEY App. B.0.
    ``means2d`` ([N,2], requires_grad) optionally receives the screen-space mean gradient
    (INRIA ``means2D.grad`` convention) for densification statistics.
    Returns (rgb [H,W,3] incl. background, depth_acc [H,W] = sum w*z, alpha [H,W], radii [N]).
    Expected depth is ``depth_acc / alpha`` (consumer side, as Splatfacto does).
    """
    need_bwd = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                               for t in (means, log_scales, quats, opac_logit, sh, means2d))
    return _Render.apply(means, log_scales, quats, opac_logit, sh, means2d, cam, sh_deg, budget, need_bwd)


# ------------------------------------------------------------------------------------------------
# gsplat-0.1 shaped operator surface
# ------------------------------------------------------------------------------------------------
def _sh_basis_torch(deg: int, d: torch.Tensor) -> torch.Tensor:
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    out = [torch.full_like(x, 0.28209479177387814)]
    if deg >= 1:
        c1 = 0.4886025119029199
        out += [-c1 * y, c1 * z, -c1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        out += [1.0925484305920792 * x * y, -1.0925484305920792 * y * z,
                0.31539156525252005 * (2 * zz - xx - yy), -1.0925484305920792 * x * z,
                0.5462742152960396 * (xx - yy)]
    if deg >= 3:
        out += [-0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * x * y * z,
                -0.4570457994644658 * y * (4 * zz - xx - yy),
                0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
                -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
                -0.5900435899266435 * x * (xx - 3 * yy)]
    return torch.stack(out, dim=-1)


class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degrees_to_use, viewdirs, coeffs):
        lib = _lib.load()
        dirs, coeffs = _f32c(viewdirs.detach()), _f32c(coeffs)
        N = coeffs.shape[0]
        out = torch.empty(N, 3, dtype=torch.float32, device=coeffs.device)
        check(lib.tgs_sh_fwd(N, degrees_to_use, coeffs.shape[1], ptr(dirs), ptr(coeffs), ptr(out), _stream()), "tgs_sh_fwd")
        ctx.deg, ctx.shape = degrees_to_use, coeffs.shape
        ctx.save_for_backward(dirs)
        return out

    @staticmethod
    def backward(ctx, v_colors):
        lib = _lib.load()
        (dirs,) = ctx.saved_tensors
        v = torch.empty(ctx.shape, dtype=torch.float32, device=dirs.device)
        check(lib.tgs_sh_bwd(ctx.shape[0], ctx.deg, ctx.shape[1], ptr(dirs), ptr(_f32c(v_colors)), ptr(v), _stream()),
              "tgs_sh_bwd")
        return None, None, v


def spherical_harmonics(degrees_to_use: int, viewdirs: torch.Tensor, coeffs: torch.Tensor):
    """gsplat-shaped SH evaluation: coeffs [N,K,3], viewdirs [N,3] (normalised in-kernel) -> [N,3]
    (no +0.5 / clamp: the caller applies them, as Splatfacto does).  Gradient flows to ``coeffs``
    only, as in gsplat 0.1.  (tgs_sh_fwd / tgs_sh_bwd; the fused training path evaluates SH inside
    K1/K8 instead.)"""
    return _SphericalHarmonics.apply(degrees_to_use, viewdirs, coeffs)


def record_xy(splats: torch.Tensor) -> torch.Tensor:
    """Absolute screen positions [N,2] from splat records: slots 0, 1 store the position relative to the
    origin (16 x0, 16 y0) of the Gaussian's packed tile rect in slot 10 (include/tgs.h; K1 evaluates
    it in compensated arithmetic, so the relative value is good to ~1e-5 px -- the sum returned here is
    rounded to fp32 like any absolute coordinate)."""
    r = splats[:, 10].contiguous().view(torch.int32)
    org = torch.stack([r & 255, (r >> 8) & 255], 1).to(torch.float32) * 16.0
    return (splats[:, 0:2] + org).contiguous()


class _ProjectGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, cam):
        means3d, quats = _f32c(means3d), _f32c(quats)
        log_scales = torch.log(_f32c(scales))
        N = means3d.shape[0]
        zero = torch.zeros(N, dtype=torch.float32, device=means3d.device)
        cam = Camera(cam.viewmat, cam.fx, cam.fy, cam.cx, cam.cy, cam.W, cam.H, cam.near,
                     cam.pix_center, cam.bg, float(glob_scale))
        splats, radii = project_fwd(cam, means3d, log_scales, quats, zero, None, -1, want_radii=True)
        ctx.cam = cam
        ctx.save_for_backward(means3d, log_scales, quats, zero, splats)
        xys = record_xy(splats)
        depths = splats[:, 2].contiguous()
        conics = splats[:, 4:7].contiguous()
        ctx.mark_non_differentiable(radii)
        return xys, depths, radii, conics

    @staticmethod
    def backward(ctx, v_xys, v_depths, _v_radii, v_conics):
        means3d, log_scales, quats, zero, splats = ctx.saved_tensors
        N = means3d.shape[0]
        v_splats = torch.zeros(N, SPLAT_FLOATS, dtype=torch.float32, device=means3d.device)
        if v_xys is not None:
            v_splats[:, 0:2] = v_xys
        if v_depths is not None:
            v_splats[:, 2] = v_depths
        if v_conics is not None:
            v_splats[:, 4:7] = v_conics
        v_means, v_ls, v_q, _, _, _ = project_bwd(ctx.cam, means3d, log_scales, quats, zero, None, -1,
                                                  splats, v_splats=v_splats)
        v_scales = v_ls / torch.exp(log_scales)
        return v_means, v_scales, None, v_q, None


def project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height,
                      img_width, block_width: int = 16, clip_thresh: float = 0.01):
    """gsplat-0.1 ``project_gaussians``: -> (xys, depths, radii, conics, comp, num_tiles_hit, cov3d).

    ``scales`` are post-exp, ``quats`` un-normalised (w,x,y,z), ``viewmat`` world->camera.
    ``comp`` is all ones (no anti-aliasing compensation in App. B); ``cov3d`` is the upper triangle
    of R S S^T R^T computed with device torch ops (not differentiated, as in gsplat 0.1).
    """
    if block_width != 16:
        raise ValueError("only 16x16 tiles are supported (SURVEY App. B.0)")
    cam = Camera(viewmat, fx, fy, cx, cy, img_width, img_height, near=clip_thresh)
    xys, depths, radii, conics = _ProjectGaussians.apply(means3d, scales, glob_scale, quats, cam)
    with torch.no_grad():
        tw, th = cam.tiles
        r = radii.to(torch.float32)
        x0 = torch.clamp(((xys[:, 0] - r) / 16).to(torch.int32), 0, tw)
        x1 = torch.clamp(((xys[:, 0] + r) / 16).to(torch.int32) + 1, 0, tw)
        y0 = torch.clamp(((xys[:, 1] - r) / 16).to(torch.int32), 0, th)
        y1 = torch.clamp(((xys[:, 1] + r) / 16).to(torch.int32) + 1, 0, th)
        num_tiles_hit = torch.where(radii > 0, (x1 - x0) * (y1 - y0), torch.zeros_like(x0))
        q = quats / quats.norm(dim=-1, keepdim=True)
        w, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)
        M = R * (scales * glob_scale)[:, None, :]
        S = M @ M.transpose(1, 2)
        cov3d = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
        comp = torch.ones_like(depths)
    return xys, depths, radii, conics, comp, num_tiles_hit, cov3d


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, colors, opacity, cam, budget):
        N = xys.shape[0]
        dev = xys.device
        op = opacity.reshape(N, 1).to(torch.float32)
        # slot 10 = packed App. B.4 tile rect x0 | y0<<8 | w<<16 | h<<24 (include/tgs.h)
        tw, th = cam.tiles
        r = radii.to(torch.float32)
        xf, yf = xys[:, 0].to(torch.float32), xys[:, 1].to(torch.float32)
        x0 = torch.clamp(((xf - r) * 0.0625).to(torch.int64), 0, tw)
        x1 = torch.clamp(((xf + r) * 0.0625).to(torch.int64) + 1, 0, tw)
        y0 = torch.clamp(((yf - r) * 0.0625).to(torch.int64), 0, th)
        y1 = torch.clamp(((yf + r) * 0.0625).to(torch.int64) + 1, 0, th)
        vis = (radii > 0) & (x1 > x0) & (y1 > y0)
        packed = torch.where(vis, x0 | (y0 << 8) | ((x1 - x0) << 16) | ((y1 - y0) << 24), torch.zeros_like(x0))
        packed = torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)
        # record slots 0, 1 hold the position relative to the origin of the tile rect (include/tgs.h)
        org = torch.stack([x0, y0], 1).to(torch.float32) * 16.0
        org = torch.where(vis[:, None], org, torch.zeros_like(org))
        splats = torch.cat([xys.to(torch.float32) - org, depths.reshape(N, 1).to(torch.float32), op,
                            conics.to(torch.float32), colors.to(torch.float32),
                            packed.reshape(N, 1).view(torch.float32),
                            torch.zeros(N, 1, dtype=torch.float32, device=dev)], dim=1).contiguous()
        group_base, tile_start, sorted_gid, _ = bin_sort(cam, splats, budget)
        rgb, depth, fT, fidx = rasterize_fwd(cam, splats, sorted_gid, tile_start)
        ctx.cam = cam
        ctx.opacity_shape = opacity.shape
        ctx.save_for_backward(splats, group_base, tile_start, sorted_gid, rgb, depth, fT, fT.stop_pos)
        return rgb, 1.0 - fT, depth

    @staticmethod
    def backward(ctx, v_rgb, v_alpha, v_depth):
        splats, group_base, tile_start, sorted_gid, rgb, depth, fT, stop = ctx.saved_tensors
        cam = ctx.cam
        partials, _ = rasterize_bwd(cam, splats, group_base, sorted_gid, tile_start, rgb, depth, fT,
                                    v_rgb, v_depth, v_alpha, stop_pos=stop)
        v = reduce_partials(cam, splats, group_base, partials)
        v_xys, v_depths = v[:, 0:2].contiguous(), v[:, 2].contiguous()
        v_op = v[:, 3].contiguous().reshape(ctx.opacity_shape)
        v_conics, v_colors = v[:, 4:7].contiguous(), v[:, 7:10].contiguous()
        return v_xys, v_depths, None, v_conics, v_colors, v_op, None, None


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                        img_width, block_width: int = 16, background=None, return_alpha: bool = False,
                        return_depth: bool = False, budget: Optional[IntersectBudget] = None):
    """gsplat-0.1 ``rasterize_gaussians``: -> out_img [H,W,3] (, out_alpha [H,W]) (, depth_acc [H,W]).

    ``num_tiles_hit`` is accepted for signature compatibility and recomputed in-kernel from
    (xys, radii) by the App. B.4 rule.  ``return_depth`` is this build's extension: the depth
    channel composited in the same pass (Splatfacto instead calls the op a second time with
    ``depths.repeat(1,3)`` as colours, which also works here).
    """
    if block_width != 16:
        raise ValueError("only 16x16 tiles are supported (SURVEY App. B.0)")
    if colors.shape[-1] != 3:
        raise ValueError("colors must be [N,3]")
    bg = (0.0, 0.0, 0.0) if background is None else tuple(float(b) for b in background.detach().cpu().tolist())
    eye = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    cam = Camera(eye, 1.0, 1.0, 0.0, 0.0, img_width, img_height, bg=bg)
    rgb, alpha, depth = _RasterizeGaussians.apply(xys, depths, radii, conics, colors, opacity, cam, budget)
    out = (rgb,)
    if return_alpha:
        out += (alpha,)
    if return_depth:
        out += (depth,)
    return out[0] if len(out) == 1 else out


# ------------------------------------------------------------------------------------------------
# image-space loss kernel (K10)
# ------------------------------------------------------------------------------------------------
def ssim_fwd_bwd(img, gt, weight: float = 1.0, want_grad: bool = True, reduce: bool = True):
    """K10 -> (sum of the SSIM map [device scalar], v_img = weight * d(sum)/d(img) or None).
    ``reduce=False`` returns the per-block partial sums instead (the train step only needs the
    total when the loss value is logged, so it skips the extra reduction launch).

    Mean SSIM = sum / (3*H*W).  For the loss term l*(1-mean SSIM) pass weight = -l/(3*H*W).
    (tgs_ssim_fwd_bwd)
    """
    lib = _lib.load()
    img, gt = _f32c(img), _f32c(gt)
    H, W = img.shape[0], img.shape[1]
    dev = img.device
    nb = ((H + 15) // 16) * ((W + 15) // 16)
    bp = torch.empty(nb, dtype=torch.float32, device=dev)
    v_img = torch.empty_like(img) if want_grad else None
    scratch = torch.empty(9 * H * W, dtype=torch.float32, device=dev) if want_grad else None
    check(lib.tgs_ssim_fwd_bwd(W, H, ptr(img), ptr(gt), C.c_float(weight), ptr(bp), ptr(v_img),
                               ptr(scratch), _stream()), "tgs_ssim_fwd_bwd")
    return (bp.sum() if reduce else bp), v_img
