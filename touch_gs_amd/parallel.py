"""Data-parallel training across views: one process per GPU, RCCL over xGMI.

The reference trains on a single GPU (no torch.distributed on any train path),
so there is no reference call pattern to follow.  Each rank holds a full replica of the flat
parameter buffer and renders a different view per iteration.  Plain form: ONE sum all-reduce of
the flat gradient buffer (59 floats per Gaussian at SH degree 3: 236 MB at 1 M), followed by the
identical fused Adam step on every rank with grad_scale = 1/world_size.  Default form
(``gather_color_reduce_geom_and_step``): the SH gradient is an outer product basis x colour
gradient, so only the 3-float colour gradient is exchanged (all-gather) and every rank rebuilds the
summed SH gradient inside the optimizer kernel; the 11 geometry gradients are all-reduced.  Either
way every rank applies the same deterministic update, so the replicas stay bit-identical.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> "GradSync":
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun) if world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TGS_DP_FORCE_COLLECTIVES=1: form the process group and issue every collective even with ONE
    # rank, so that the RCCL code path (all_gather_into_tensor / all_reduce on the side stream) can be
    # exercised on a single-GPU box.
    force = os.environ.get("TGS_DP_FORCE_COLLECTIVES", "") not in ("", "0")
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # "nccl" IS RCCL on ROCm.  TGS_DIST_BACKEND=gloo lets the multi-rank code path be
            # exercised with several ranks on ONE GPU (RCCL refuses duplicate devices).
            backend = os.environ.get("TGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return GradSync(rank, world, local, force_collectives=force)


class _RawDeviceArray:
    """__cuda_array_interface__ view of raw device memory (IPC-mapped peer buffers) for torch.as_tensor."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3}


class PeerExchange:
    """The factored gradient exchange by direct stores into IPC-mapped peer memory (csrc/peer.hip; SURVEY section 5's
    alternative to RCCL: xGMI is point-to-point, so every rank writes to its 7 peers at once, one hop, no collective).

    Every rank owns ONE receive buffer (uncached device memory, zeroed), mapped into all other processes:

        per parity (2):  colour[C][world][chunk_c] | rs[world][slice] | ag[world * slice]        (floats)
        then:            flags colour[C][world] | rs[world] | ag[world] | err | tickets[C + 2]    (int32)

    * all-gather of chunk c's colour block: the sender stores it into slot `rank` of colour[c] on every rank (itself
      included) and raises flag colour[c][rank] there; the SH Adam of chunk c waits for the `world` flags of its own buffer;
    * all-reduce of the geometry gradients = reduce-scatter + all-gather: slice q of the local gradient goes to rank
      q's rs[rank]; rank q sums its `world` received slices IN RANK ORDER and stores the result into ag[q * slice] on
      every rank -- each element is summed once, by one rank, so all replicas receive the same bits.
    Data slots alternate with the parity of the exchange's sequence number; a sender cannot run two exchanges ahead
    of a receiver (each exchange ends with flags from every peer), so a slot is never overwritten while it is read.
    Bit-identical to the collective form for the colour part (a copy) and for two ranks (a + b); with more ranks the
    rank-order sum differs from a ring's association by rounding -- the replicas stay identical either way."""

    def __init__(self, dp: "GradSync", device, chunk_floats: Sequence[int], geom_floats: int):
        import ctypes as C
        from . import _lib
        self.lib = lib = _lib.load()
        self.W, self.r, self.dev = dp.world, dp.rank, device
        W = self.W
        if W > 8:
            raise ValueError("the peer transport addresses at most 8 ranks (one node)")
        self.chunk = [int(m) for m in chunk_floats]
        self.C = len(self.chunk)
        self.G = int(geom_floats)
        self.slice = -(-(-(-self.G // W)) // 4) * 4
        pad = lambda n: -(-n // 4) * 4
        self.col_off, off = [], 0
        for m in self.chunk:
            self.col_off.append(off)
            off += pad(W * m)
        self.rs_off, off = off, off + W * self.slice
        self.ag_off, off = off, off + W * self.slice
        self.parity_floats = pad(off)
        self.n_float = 2 * self.parity_floats
        self.f_col, self.f_rs, self.f_ag = 0, self.C * W, self.C * W + W
        self.f_err = self.C * W + 2 * W
        self.f_ticket = self.f_err + 1
        self.n_int = -(-(self.f_ticket + self.C + 2) // 4) * 4
        nbytes = 4 * (self.n_float + self.n_int)
        base, handle = C.c_void_p(), (C.c_ubyte * 64)()
        # uncached or fine-grained memory only (TGS_PEER_MEM_UNCACHED | _FINEGRAINED): with plain cached memory the
        # owner's L2 may hold stale lines of a slot a peer has written and the premise of the transport is void --
        # the call fails instead of downgrading; the kind obtained goes into the bench line (dp_exchange.memory_kind)
        kind = C.c_int(0)
        _lib.check(lib.tgs_peer_alloc(nbytes, 1 | 2, C.byref(base), handle, C.byref(kind)), "tgs_peer_alloc")
        self.memory_kind = {1: "uncached", 2: "fine-grained"}[kind.value]
        self._own = base.value
        handles = [None] * W
        dist.all_gather_object(handles, bytes(handle))
        self.base = []
        for q in range(W):
            if q == self.r:
                self.base.append(self._own)
            else:
                p, h = C.c_void_p(), (C.c_ubyte * 64).from_buffer_copy(handles[q])
                _lib.check(lib.tgs_peer_open(h, C.byref(p)), "tgs_peer_open")
                self.base.append(p.value)
        self.seq = 0
        self.bytes_pushed = 0
        # device words a timed-out wait sets to 1 (tgs_peer_wait): the sticky overflow word and word [1] of the agreed
        # verdict the optimizer kernels are guarded by -- the kernels behind the wait then skip instead of consuming
        # stale receive slots (GradSync.set_poison_words; None = no guard words exist, e.g. a synchronous budget)
        self.poison = (None, None)
        self.timeout_s = float(os.environ.get("TGS_PEER_TIMEOUT_S", "0") or 0)    # <= 0: the library's 20 s
        # default: the data kernels publish nothing, a separate one-wave launch raises the flags behind the kernel
        # boundary -- ordered by the stream alone.  TGS_PEER_SAFE_FLAGS=0: the last workgroup of the data kernel
        # raises them itself (one launch less per transfer); that form assumes that a completed store to uncached
        # memory has arrived at its destination (csrc/peer.hip), which no run has exercised across GPUs
        self.safe_flags = os.environ.get("TGS_PEER_SAFE_FLAGS", "1") not in ("", "0")
        # everything a step needs is built ONCE: tensor views of the receive slots (torch.as_tensor on a raw pointer
        # queries the pointer's attributes: ~0.3 ms of host time each) and the host arrays of peer addresses
        A, r = self._arr, self.r
        self._views = {(p_, c): torch.as_tensor(_RawDeviceArray(self._f(r, p_, self.col_off[c]), W * self.chunk[c], "<f4"),
                                                device=device).view(W, self.chunk[c])
                       for p_ in (0, 1) for c in range(self.C)}
        self._geom_views = [torch.as_tensor(_RawDeviceArray(self._f(r, p_, self.ag_off), self.G, "<f4"), device=device)
                            for p_ in (0, 1)]
        self._err_view = torch.as_tensor(_RawDeviceArray(self._i(r, self.f_err), 1, "<i4"), device=device)
        sl = self.slice
        self._a_col = {(p_, c): (A([self._f(q, p_, self.col_off[c] + r * self.chunk[c]) for q in range(W)]),
                                 A([self._i(q, self.f_col + c * W + r) for q in range(W)]))
                       for p_ in (0, 1) for c in range(self.C)}
        self._a_col_wait = [A([self._i(r, self.f_col + c * W + s) for s in range(W)]) for c in range(self.C)]
        self._a_rs = [(A([self._f(q, p_, self.rs_off + r * sl) for q in range(W)]), A([self._i(q, self.f_rs + r) for q in range(W)]))
                      for p_ in (0, 1)]
        self._a_rs_wait = A([self._i(r, self.f_rs + q) for q in range(W)])
        self._a_red = [(A([self._f(r, p_, self.rs_off + q * sl) for q in range(W)]),
                        A([self._f(q, p_, self.ag_off + r * sl) for q in range(W)]),
                        A([self._i(q, self.f_ag + r) for q in range(W)])) for p_ in (0, 1)]
        self._a_ag_wait = A([self._i(r, self.f_ag + q) for q in range(W)])
        dist.barrier()            # every mapping exists before the first store

    # -- addresses ---------------------------------------------------------------------------------------------
    def _f(self, q: int, parity: int, off: int) -> int:
        return self.base[q] + 4 * (parity * self.parity_floats + off)

    def _i(self, q: int, idx: int) -> int:
        return self.base[q] + 4 * (self.n_float + idx)

    @staticmethod
    def _arr(vals):
        import ctypes as C
        return (C.c_void_p * len(vals))(*vals)

    def colour_all(self, parity: int, c: int) -> torch.Tensor:
        """This rank's receive slots of chunk c as a [world, chunk_c] tensor (what the SH Adam reads)."""
        return self._views[(parity, c)]

    def geom_reduced(self, parity: int) -> torch.Tensor:
        return self._geom_views[parity]

    # -- operations (enqueued on the current stream) -------------------------------------------------------------
    def begin(self) -> int:
        self.seq += 1
        return self.seq & 1

    def push_colour(self, c: int, block: torch.Tensor) -> None:
        from . import _lib
        W, r, p, m = self.W, self.r, self.seq & 1, self.chunk[c]
        dsts, flags = self._a_col[(p, c)]
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.tgs_peer_push(W, dsts, None if self.safe_flags else flags, _lib.ptr(block), 4 * m, self.seq,
                                          self._i(r, self.f_ticket + c), s), "tgs_peer_push")
        self._signal(flags, s)
        self.bytes_pushed += 4 * m * (W - 1)

    def _signal(self, flags, stream) -> None:
        if self.safe_flags:
            from . import _lib
            _lib.check(self.lib.tgs_peer_signal(self.W, flags, self.seq, stream), "tgs_peer_signal")

    def wait_colour(self, c: int) -> None:
        self._wait(self._a_col_wait[c])

    def _wait(self, flags) -> None:
        from . import _lib
        _lib.check(self.lib.tgs_peer_wait(len(flags), flags, self.seq, self._i(self.r, self.f_err), self.timeout_s,
                                          self.poison[0], self.poison[1],
                                          torch.cuda.current_stream().cuda_stream), "tgs_peer_wait")

    def all_reduce_geom(self, geom_grad: torch.Tensor) -> None:
        """reduce-scatter (direct) -> rank-order sum of the own slice -> all-gather (direct); comm stream."""
        from . import _lib
        W, r, p, sl = self.W, self.r, self.seq & 1, self.slice
        s = torch.cuda.current_stream().cuda_stream
        dsts, flags = self._a_rs[p]
        _lib.check(self.lib.tgs_peer_scatter(W, dsts, None if self.safe_flags else flags, _lib.ptr(geom_grad), 4 * sl, 4 * self.G,
                                             self.seq, self._i(r, self.f_ticket + self.C), s), "tgs_peer_scatter")
        self._signal(flags, s)
        self._wait(self._a_rs_wait)
        mine = max(0, min(sl, self.G - r * sl))
        srcs, dsts, flags = self._a_red[p]
        _lib.check(self.lib.tgs_peer_reduce_push(W, srcs, W, dsts, None if self.safe_flags else flags, 4 * mine, self.seq,
                                                 self._i(r, self.f_ticket + self.C + 1), s), "tgs_peer_reduce_push")
        self._signal(flags, s)
        self.bytes_pushed += 4 * (self.G - mine) + 4 * mine * (W - 1)

    def wait_geom(self) -> None:
        self._wait(self._a_ag_wait)

    def check(self) -> None:
        """Raises if a wait timed out (a peer never delivered): synchronises the device."""
        torch.cuda.synchronize(self.dev)
        self.raise_if(int(self._err_view.item()))

    def raise_if(self, e: int) -> None:
        """``e`` = a host copy of the error word (``err_word`` rides in the trainer's per-step status copy)."""
        if e:
            raise RuntimeError(f"peer exchange: rank {self.r} timed out waiting for the flag of rank {e - 1}; the steps "
                               "behind the timed-out wait were voided on the device (poison words), nothing stale was applied")

    @property
    def err_word(self) -> torch.Tensor:
        return self._err_view

    def close(self) -> None:
        from . import _lib
        if self.base is None:
            return
        torch.cuda.synchronize(self.dev)
        dist.barrier()            # nobody stores into a buffer that is about to be unmapped
        for q, b in enumerate(self.base):
            if q != self.r:
                self.lib.tgs_peer_close(b)
        dist.barrier()
        self.lib.tgs_peer_free(self._own)
        self.base = None


class GradSync:
    """Gradient exchange for data-parallel training.

    ``all_reduce_`` is the plain form (one blocking sum all-reduce, returns the 1/world scale).
    ``reduce_and_step`` is the pipelined form used by the trainer: the flat gradient buffer is cut
    into ``n_chunks`` contiguous ranges; their all-reduces are enqueued back to back on a side
    stream and the fused Adam of range c is enqueued on the compute stream behind an event that
    fires when range c has been reduced -- so the optimizer (~0.3 ms at 1 M Gaussians) runs under
    the shadow of the collective (estimated >1 ms for 236 MB on xGMI) instead of after it.
    ``gather_color_reduce_geom_and_step`` is the factored exchange in one piece;
    ``pipelined_color_exchange_and_step`` -- the default of the trainer -- is the same exchange cut into row
    chunks, K8 included (see its docstring).
    """

    def __init__(self, rank: int = 0, world: int = 1, local_rank: int = 0, n_chunks: int = 8,
                 force_collectives: bool = False, color_chunks: Optional[int] = None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.n_chunks = n_chunks
        # row chunks of the pipelined factored exchange (1 = one K8, one gather).  Every chunk costs ~16 us of
        # launches and workgroup-round tails (measured with one rank, where nothing is hidden: 1.145 / 1.209 /
        # 1.307 ms per step with 1 / 4 / 8 chunks); what it buys is link time hidden behind K8 and the SH Adam, which
        # grows with the number of ranks -- so: 4 chunks from 4 ranks on, one piece below.  TGS_DP_COLOR_CHUNKS overrides.
        env = os.environ.get("TGS_DP_COLOR_CHUNKS")
        self.color_chunks = color_chunks if color_chunks is not None else int(env) if env else (4 if world >= 4 else 1)
        # collectives are issued when there is more than one rank -- or when forced (1-rank RCCL test)
        self.active = world > 1 or (force_collectives and dist.is_initialized())
        self.bytes_per_step = 0
        self._comm_stream = None
        self.timing = False        # record events around the collectives of the factored step
        self._comm_events = None
        # transport of the factored exchange: "rccl" = torch.distributed collectives (default), "ipc" = direct stores
        # into IPC-mapped peer buffers (PeerExchange; one node, <= 8 ranks).  TGS_DP_TRANSPORT selects it.
        self.transport = os.environ.get("TGS_DP_TRANSPORT", "rccl").lower()
        self.peer = None

    def all_reduce_(self, flat_grad: torch.Tensor) -> float:
        if self.active:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
            self.bytes_per_step = flat_grad.numel() * flat_grad.element_size()
        return 1.0 / self.world

    @staticmethod
    def chunk_ranges(numel: int, n_chunks: int, align: int = 4):
        """Contiguous [begin, end) ranges covering [0, numel), every boundary a multiple of `align`."""
        n_chunks = max(1, min(n_chunks, max(numel // align, 1)))
        per = -(-numel // n_chunks)
        per = -(-per // align) * align
        out, b = [], 0
        while b < numel:
            e = min(b + per, numel)
            out.append((b, e))
            b = e
        return out

    def reduce_and_step(self, flat_grad: torch.Tensor, step_range, begin_step=None) -> None:
        """Sum-all-reduce ``flat_grad`` chunk by chunk and call ``step_range(begin, end, scale)`` for
        every chunk once it is reduced (scale = 1/world turns the sum into a mean)."""
        scale = 1.0 / self.world
        if begin_step is not None:
            begin_step()
        if not self.active:
            step_range(0, flat_grad.numel(), 1.0)
            return
        ranges = self.chunk_ranges(flat_grad.numel(), self.n_chunks)
        self.bytes_per_step = flat_grad.numel() * flat_grad.element_size()
        if not flat_grad.is_cuda:  # CPU / gloo (tests): no streams, same order of operations
            for b, e in ranges:
                dist.all_reduce(flat_grad[b:e], op=dist.ReduceOp.SUM)
                step_range(b, e, scale)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=flat_grad.device)
        comp = torch.cuda.current_stream(flat_grad.device)
        ready = torch.cuda.Event()
        ready.record(comp)                     # gradients complete on the compute stream
        self._comm_stream.wait_event(ready)
        events = []
        with torch.cuda.stream(self._comm_stream):
            for b, e in ranges:
                dist.all_reduce(flat_grad[b:e], op=dist.ReduceOp.SUM)
                ev = torch.cuda.Event()
                ev.record(self._comm_stream)
                events.append(ev)
        for (b, e), ev in zip(ranges, events):
            comp.wait_event(ev)
            step_range(b, e, scale)

    def gather_color_reduce_geom_and_step(self, geom_grad: torch.Tensor, v_color: torch.Tensor,
                                          v_color_all: torch.Tensor, step_sh, step_geom, begin_step=None) -> None:
        """Factored exchange of one data-parallel step (SH gradient = basis x colour gradient):

        * all-gather every rank's colour-gradient block ``v_color`` [3N+4] into ``v_color_all``
          [world, 3N+4], then ``step_sh(v_color_all, 1/world)`` (Adam on the SH segment; rebuilds
          sum_r Y(dir_r) v_color_r itself);
        * sum-all-reduce the geometry gradients ``geom_grad`` (11 floats per Gaussian), then
          ``step_geom(0, geom_grad.numel(), 1/world)``.

        Both collectives run back to back on a side stream; the SH update (the bulk of the optimizer)
        overlaps the geometry all-reduce.  Per Gaussian 12*world + 44 B cross the fabric instead of
        4*(11+3K) (236 B at SH degree 3)."""
        scale = 1.0 / self.world
        if begin_step is not None:
            begin_step()
        gather = (lambda: v_color_all.copy_(v_color.view(1, -1))) if not self.active else \
                 (lambda: dist.all_gather_into_tensor(v_color_all.view(-1), v_color)) if dist.get_backend() == "nccl" else \
                 (lambda: dist.all_gather(list(v_color_all.view(self.world, -1).unbind(0)), v_color))
        self.bytes_per_step = 4 * (v_color.numel() * self.world + geom_grad.numel())
        if not self.active or not geom_grad.is_cuda:
            gather()
            step_sh(v_color_all, scale)
            if self.active:
                dist.all_reduce(geom_grad, op=dist.ReduceOp.SUM)
            step_geom(0, geom_grad.numel(), scale)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=geom_grad.device)
        comp = torch.cuda.current_stream(geom_grad.device)
        ready = torch.cuda.Event()
        ready.record(comp)                     # K8 complete on the compute stream
        self._comm_stream.wait_event(ready)
        t = self.timing
        with torch.cuda.stream(self._comm_stream):
            begun = torch.cuda.Event(enable_timing=t)
            begun.record(self._comm_stream)
            gather()
            gathered = torch.cuda.Event(enable_timing=t)
            gathered.record(self._comm_stream)
            dist.all_reduce(geom_grad, op=dist.ReduceOp.SUM)
            reduced = torch.cuda.Event(enable_timing=t)
            reduced.record(self._comm_stream)
        if t:
            self._comm_events = (begun, gathered, reduced, v_color.numel() * 4, geom_grad.numel() * 4)
        comp.wait_event(gathered)
        step_sh(v_color_all, scale)
        comp.wait_event(reduced)
        step_geom(0, geom_grad.numel(), scale)

    def fused_tail(self) -> bool:
        """Run the tail of a data-parallel step (SH Adam of every chunk + geometry Adam + the next view's K1) as ONE launch
        behind the geometry all-reduce (optim.step_sh_gathered_geom_and_project_next) instead of chunk by chunk?
        ``TGS_DP_FUSED_TAIL`` = 1 / 0 forces it; default (auto): only for a one-rank group, where no transfer exists for
        the chunked SH Adam to hide under -- measured with 1, 2 and 4 ranks on one GPU (profiles/r6_dp_fused_tail.json:
        the fused launch saves launches and one pass over the SH rows, but every rank then waits for the whole exchange
        before it touches the SH rows)."""
        mode = os.environ.get("TGS_DP_FUSED_TAIL", "auto")
        if mode in ("0", "1"):
            return mode == "1"
        return self.world == 1

    def color_chunk_rows(self, N: int, align: int = 256):
        """Row ranges [begin, end) of the pipelined exchange: at most ``color_chunks`` ranges covering [0, N),
        every boundary a multiple of ``align`` (K8's workgroup / the binning group: 256 rows)."""
        n = max(1, min(self.color_chunks, N // align))
        per = -(-N // n)
        per = -(-per // align) * align
        out, b = [], 0
        while b < N:
            out.append((b, min(b + per, N)))
            b += per
        # never empty: chunk 0 carries the overflow verdict (dp_agree_overflow), so a model without Gaussians still
        # exchanges one (empty) block
        return out or [(0, 0)]

    def pipelined_color_exchange_and_step(self, geom_grad: torch.Tensor, blocks: Sequence[torch.Tensor],
                                          blocks_all: Sequence[torch.Tensor], backward_chunk, step_sh_chunk,
                                          step_geom, begin_step=None) -> None:
        """The factored exchange of ``gather_color_reduce_geom_and_step`` pipelined over row chunks, K8 included:

        compute stream:  K8(0) K8(1) ... K8(C-1)   SH-Adam(0) ... SH-Adam(C-1)              geometry Adam
        side stream:           gather(0) gather(1) ... gather(C-1)  all-reduce(geometry gradients)

        ``backward_chunk(c)`` launches K8 in colour mode for chunk c (geometry gradients of its rows into
        ``geom_grad``, its colour block into ``blocks[c]``); chunk c's block is all-gathered into
        ``blocks_all[c]`` [world, len(blocks[c])] as soon as its K8 has run -- while the later chunks' K8 still
        runs -- and ``step_sh_chunk(c, blocks_all[c], 1/world)`` (Adam on the SH rows of the chunk) starts when
        its gather has landed, under the later gathers and the geometry all-reduce.  Exposed link time: the
        gathers minus the tail of K8, the all-reduce minus the last chunk's SH Adam.  Same arithmetic per row
        as the unchunked form, so the results are bit-identical to it."""
        scale = 1.0 / self.world
        C_ = len(blocks)
        if begin_step is not None:
            begin_step()
        nccl = self.active and dist.get_backend() == "nccl"

        def gather(c):
            if not self.active:
                blocks_all[c].copy_(blocks[c].view(1, -1))
            elif nccl:
                dist.all_gather_into_tensor(blocks_all[c].view(-1), blocks[c])
            else:
                dist.all_gather(list(blocks_all[c].view(self.world, -1).unbind(0)), blocks[c])

        self.bytes_per_step = 4 * (sum(b.numel() for b in blocks) * self.world + geom_grad.numel())
        if self.transport == "ipc" and self.active and geom_grad.is_cuda:
            return self._peer_exchange_and_step(geom_grad, blocks, blocks_all, backward_chunk, step_sh_chunk, step_geom, scale)
        if not self.active or not geom_grad.is_cuda:
            for c in range(C_):
                backward_chunk(c)
                gather(c)
            for c in range(C_):
                step_sh_chunk(c, blocks_all[c], scale)
            if self.active:
                dist.all_reduce(geom_grad, op=dist.ReduceOp.SUM)
            step_geom(0, geom_grad.numel(), scale)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=geom_grad.device)
        comp, comm = torch.cuda.current_stream(geom_grad.device), self._comm_stream
        t = self.timing
        gathered = []
        begun = None
        for c in range(C_):
            backward_chunk(c)
            done = torch.cuda.Event()
            done.record(comp)                  # chunk c's block (and geometry gradients) complete
            comm.wait_event(done)
            with torch.cuda.stream(comm):
                if c == 0:
                    begun = torch.cuda.Event(enable_timing=t)
                    begun.record(comm)
                gather(c)
                ev = torch.cuda.Event(enable_timing=t and c == C_ - 1)
                ev.record(comm)
                gathered.append(ev)
        with torch.cuda.stream(comm):          # behind the last chunk's K8: every geometry gradient is there
            dist.all_reduce(geom_grad, op=dist.ReduceOp.SUM)
            reduced = torch.cuda.Event(enable_timing=t)
            reduced.record(comm)
        if t:
            self._comm_events = (begun, gathered[-1], reduced, sum(b.numel() for b in blocks) * 4, geom_grad.numel() * 4)
        for c in range(C_):
            comp.wait_event(gathered[c])
            step_sh_chunk(c, blocks_all[c], scale)
        comp.wait_event(reduced)
        step_geom(0, geom_grad.numel(), scale)

    def _peer_exchange_and_step(self, geom_grad, blocks, blocks_all, backward_chunk, step_sh_chunk, step_geom, scale) -> None:
        """The pipelined exchange over the peer transport: same schedule, the collectives replaced by direct stores.

        compute stream:  K8(0) ... K8(C-1)   [wait flags 0] SH-Adam(0) ... [wait flags C-1] SH-Adam(C-1)   [wait ag] copy, geometry Adam
        side stream:         push(0) ... push(C-1)   scatter slices -> [wait rs] rank-order sum + push to all"""
        # ``blocks_all`` are the caller's ordinary (cached) buffers: a received chunk is copied out of the uncached
        # receive slots by one vectorised copy before the SH Adam reads it with 4-byte loads (which crawl on uncached
        # memory: 759 instead of 200 us)
        sizes = [int(b.numel()) for b in blocks]
        if self.peer is None or self.peer.chunk != sizes or self.peer.G != geom_grad.numel():
            if self.peer is not None:
                self.peer.close()
            self.peer = PeerExchange(self, geom_grad.device, sizes, geom_grad.numel())
            self.peer.poison = getattr(self, "_poison", (None, None))
        peer, C_ = self.peer, len(blocks)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=geom_grad.device)
        comp, comm = torch.cuda.current_stream(geom_grad.device), self._comm_stream
        parity = peer.begin()
        t = self.timing
        begun = gathered = None
        for c in range(C_):
            backward_chunk(c)
            done = torch.cuda.Event()
            done.record(comp)
            comm.wait_event(done)
            with torch.cuda.stream(comm):
                if c == 0:
                    begun = torch.cuda.Event(enable_timing=t)
                    begun.record(comm)
                peer.push_colour(c, blocks[c])
        with torch.cuda.stream(comm):
            gathered = torch.cuda.Event(enable_timing=t)
            gathered.record(comm)
            peer.all_reduce_geom(geom_grad)
            reduced = torch.cuda.Event(enable_timing=t)
            reduced.record(comm)
        if t:
            self._comm_events = (begun, gathered, reduced, sum(sizes) * 4, geom_grad.numel() * 4)
        for c in range(C_):
            peer.wait_colour(c)
            blocks_all[c].copy_(peer.colour_all(parity, c))
            step_sh_chunk(c, blocks_all[c], scale)
        peer.wait_geom()
        geom_grad.copy_(peer.geom_reduced(parity))
        step_geom(0, geom_grad.numel(), scale)

    def check_transport(self) -> None:
        """Raises if a peer-exchange wait timed out (synchronises); no-op for the collective transport."""
        if self.peer is not None:
            self.peer.check()

    def set_poison_words(self, sticky, verdict) -> None:
        """The device words a timed-out peer wait raises (``tgs_peer_wait``): the budget's sticky overflow word and the
        agreed verdict ``int32[2]`` whose word [1] guards the optimizer kernels.  Kept on the GradSync so that a
        PeerExchange built later (first step) picks them up."""
        from . import _lib
        self._poison = (_lib.ptr(sticky) if sticky is not None else None,
                        (_lib.ptr(verdict) + 4) if verdict is not None else None)
        if self.peer is not None:
            self.peer.poison = self._poison

    def comm_report(self) -> Optional[dict]:
        """Times and bus bandwidths of the last factored exchange recorded with ``timing = True``
        (synchronises).  busbw follows the rccl-tests convention: all-gather total_bytes*(n-1)/n / t,
        all-reduce bytes*2(n-1)/n / t."""
        if self._comm_events is None:
            return None
        begun, gathered, reduced, block_bytes, geom_bytes = self._comm_events
        reduced.synchronize()
        n = self.world
        tg, tr = begun.elapsed_time(gathered) * 1e-3, gathered.elapsed_time(reduced) * 1e-3
        return {"all_gather_ms": round(tg * 1e3, 4), "all_gather_bytes_per_rank": block_bytes,
                "all_gather_busbw_GBs": round(block_bytes * n * (n - 1) / n / tg / 1e9, 3),
                "all_reduce_ms": round(tr * 1e3, 4), "all_reduce_bytes": geom_bytes,
                "all_reduce_busbw_GBs": round(geom_bytes * 2 * (n - 1) / n / tr / 1e9, 3)}

    def busbw_sweep(self, device, sizes_mb=(12, 44, 96, 236), reps: int = 5) -> Optional[dict]:
        """rccl-tests style message-size sweep on the job's own process group: for every size S the
        time and bus bandwidth of an all-gather of S per rank, an all-reduce of an S-byte buffer, and -- peer
        transport -- a push of S to every peer + flags.  busbw: all-gather S (n-1) / t (per-rank receive rate),
        all-reduce 2 S (n-1)/n / t, push S (n-1) / t.  12 / 44 MB are the colour block and the geometry gradient of the
        1 M-Gaussian step, 96 / 236 MB the gathered block and the dense gradient.  Returns None with one rank."""
        if not self.active or self.world < 2:
            return None
        n, out = self.world, {}
        nccl = dist.get_backend() == "nccl"
        if not nccl:        # gloo (tests: ranks sharing one GPU, payloads through host memory): a token sweep
            sizes_mb, reps = tuple(sizes_mb)[:2], 1
        peer = None
        for mb in sizes_mb:
            m = mb * 1_000_000 // 16 * 4                       # floats, 16-byte multiple
            src = torch.ones(m, dtype=torch.float32, device=device)
            allb = torch.empty(n * m, dtype=torch.float32, device=device)

            def gather():
                if nccl:
                    dist.all_gather_into_tensor(allb, src)
                else:
                    dist.all_gather(list(allb.view(n, m).unbind(0)), src)

            ops_ = {"all_gather": gather, "all_reduce": lambda: dist.all_reduce(src, op=dist.ReduceOp.SUM)}
            if self.transport == "ipc":
                if peer is not None:
                    peer.close()
                peer = PeerExchange(self, device, [m], 4)

                def push():
                    peer.begin()
                    peer.push_colour(0, src)
                    peer.wait_colour(0)
                ops_["peer_push"] = push
            row = {}
            for name, fn in ops_.items():
                for _ in range(2):
                    fn()
                torch.cuda.synchronize(device)
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                e1.synchronize()
                t = self.max_over_ranks(e0.elapsed_time(e1) / reps * 1e-3)
                S = 4 * m
                bw = (2 * S * (n - 1) / n if name == "all_reduce" else S * (n - 1)) / t / 1e9
                row[name] = {"ms": round(t * 1e3, 4), "busbw_GBs": round(bw, 2)}
            out[f"{mb}MB"] = row
            del src, allb
        if peer is not None:
            peer.check()
            peer.close()
        return out

    def barrier(self):
        if self.active:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return value
        dev = torch.device("cuda", self.local_rank) if (dist.get_backend() == "nccl") else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def views_for_step(self, step: int, n_views: int) -> int:
        """Rank r renders view (step*world + r) mod n_views: every iteration covers `world`
        distinct consecutive views, every rank sees every view over time."""
        return (step * self.world + self.rank) % n_views

    def assert_replicas_identical(self, flat_params: torch.Tensor):
        """Cheap divergence check: max over ranks of a checksum must equal the local one."""
        if not self.active:
            return
        s = flat_params.double().sum().reshape(1)
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if float(lo) != float(hi):
            raise RuntimeError("data-parallel replicas diverged")


def split_train_indices(n: int, fraction: float):
    """Train/eval split rule of the reference (utils/create_point_cloud_from_touches.py:174-198):
    i_train = linspace(0, n-1, ceil(n*f)+1, dtype=int)[:-1] (truncation), i_eval = the rest.
    Returns (i_train, i_eval) as lists; like the reference it refuses splits whose truncated
    indices collide."""
    import math
    import numpy as np
    num_train = math.ceil(n * fraction)
    i_all = np.arange(n)
    i_train = np.linspace(0, n - 1, num_train + 1, dtype=int)[:-1]
    i_eval = np.setdiff1d(i_all, i_train)
    if len(i_eval) != n - num_train:
        raise ValueError(f"train split fraction {fraction} of {n} images yields duplicate indices")
    return [int(v) for v in i_train], [int(v) for v in i_eval]
