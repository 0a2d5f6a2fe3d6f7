"""Splatfacto-style Gaussian model for the ``depth-gaussian-splatting`` method of Touch-GS.

The reference trains through ``ns-train depth-gaussian-splatting --pipeline.model.depth-loss-mult ..
--pipeline.model.depth-loss-type {DEPTH_UNCERTAINTY_WEIGHTED_LOSS,SIMPLE_LOSS}
--pipeline.model.uncertainty_weight ..`` (reference scripts/train_bunny_real.sh:52,
train_block_data.sh:50, train_bunny_blender.sh:50); the model class itself lives in an absent
submodule, so the method surface mirrors the in-tree nerfstudio plugin evidence
(legacy/model_tactile.py:68,77,103,138,192: populate_modules / get_outputs / get_metrics_dict /
get_loss_dict / get_image_metrics_and_images) and the loss follows SURVEY section 8 row a11.

Two execution paths over the same kernels:
* ``get_outputs`` + ``get_loss_dict`` -- autograd path (ops.render is a torch.autograd.Function);
* ``train_step``                      -- fused path: forward, SSIM, compositing backward with the
  L1 + tactile depth/uncertainty loss evaluated in-kernel, projection backward straight into the
  flat gradient buffer, optional RCCL all-reduce, fused Adam.  No autograd graph, no host sync.
"""
from __future__ import annotations

import collections
import os
import dataclasses
import math
from typing import Dict, Optional

import torch

from . import ops
from .camera import Camera
from .optim import FusedAdam, GaussianParams

DEPTH_LOSS_TYPES = ("DEPTH_UNCERTAINTY_WEIGHTED_LOSS", "SIMPLE_LOSS")


@dataclasses.dataclass
class ModelConfig:
    """Flag names follow the reference's tyro flags (scripts/train_*.sh) and Splatfacto defaults
    (SURVEY App. A.3)."""
    sh_degree: int = 3
    ssim_lambda: float = 0.2
    # run SSIM on a second stream, pipelined by image bands behind K7 (ops.rasterize_bwd_ssim_pipelined): same
    # results.  OFF by default -- measured slower: a band launch of K7 no longer fills the GPU (1080p: four
    # launches of 2040 one-wave tiles take 4 x 210 us against 450 us for the whole image; 4K: 4.60 against
    # 4.43 ms per step), although SSIM itself hides almost completely under K7
    pipeline_ssim: bool = False
    pipeline_ssim_min_tiles: int = 0
    depth_loss_mult: float = 0.2          # scripts/train_block_data.sh:50
    depth_loss_type: str = "DEPTH_UNCERTAINTY_WEIGHTED_LOSS"
    uncertainty_weight: float = 1.0
    depth_eps: float = 1e-6
    sh_degree_interval: int = 1000
    # Splatfacto's coarse-to-fine schedule (num_downscales 2, resolution_schedule 250): training
    # starts on images downscaled by 2^num_downscales and doubles the resolution every resolution_schedule steps.
    # 0 = always full resolution (the default of this library's ops and of bench.py; the trainers set 2 / 250)
    num_downscales: int = 0
    resolution_schedule: int = 250
    background_color: tuple = (0.0, 0.0, 0.0)
    # Adam learning rates per group (Splatfacto defaults)
    lr_means: float = 1.6e-4
    # Splatfacto's position learning rate decays exponentially 1.6e-4 -> 1.6e-6 over the run
    # (UNVERIFIED-PRIOR); lr_means_final = None keeps it constant
    lr_means_final: Optional[float] = 1.6e-6
    lr_means_max_steps: int = 30000
    lr_scales: float = 5e-3
    lr_quats: float = 1e-3
    lr_opac: float = 5e-2
    lr_sh_dc: float = 2.5e-3
    lr_sh_rest: float = 1.25e-4
    # keep the Gaussians in 3-D Morton order (model.spatial_sort() after every densification); the
    # trainer and bench.py switch it on, library users keep their own row order by default
    spatial_sort: bool = False
    # full Morton re-sort every this many refinements (in between, refine() keeps children next to parents)
    resort_every_refines: int = 10
    # spatial_sort deals the Gaussians that covered more than 32 tiles in the last frame evenly over the binning groups
    # (optim.balanced_order): a layout choice like the Morton order itself, results unchanged
    balance_long_runs: bool = True
    long_run: int = 0          # tiles beyond which a Gaussian is a long run (tgs_set_long_run); 0 = chosen at every re-sort (32, or 8 for object-centric frames)

    def downscale_factor(self, step: int) -> int:
        """2 ** max(num_downscales - step // resolution_schedule, 0)  (Splatfacto._get_downscale_factor)."""
        if self.num_downscales <= 0 or self.resolution_schedule <= 0:
            return 1
        return 2 ** max(self.num_downscales - step // self.resolution_schedule, 0)

    def lr_means_at(self, step: int) -> float:
        """ExponentialDecay schedule: lr_init * (lr_final / lr_init) ** min(step / max_steps, 1)."""
        if self.lr_means_final is None or self.lr_means_max_steps <= 0:
            return self.lr_means
        t = min(max(step, 0) / self.lr_means_max_steps, 1.0)
        return float(math.exp(math.log(self.lr_means) * (1 - t) + math.log(self.lr_means_final) * t))

    def lrs(self) -> Dict[str, float]:
        return dict(means=self.lr_means, log_scales=self.lr_scales, quats=self.lr_quats,
                    opac_logit=self.lr_opac, sh_dc=self.lr_sh_dc, sh_rest=self.lr_sh_rest)


@dataclasses.dataclass
class View:
    """One training view: camera + supervision images on the device.

    ``depth`` (metres x dataparser scale, 0 = unsupervised) and ``uncertainty`` are what the
    reference's plumbing writes as 16-bit mm PNGs (utils/fuse_touch_vision.py:372-376) and
    registers in transforms.json (utils/add_depth_file_path_to_transforms.py:37-50).
    """
    cam: Camera
    rgb: torch.Tensor                      # [H,W,3] in [0,1]
    depth: Optional[torch.Tensor] = None   # [H,W]
    uncertainty: Optional[torch.Tensor] = None  # [H,W]
    n_valid_depth: Optional[int] = None    # cached count of depth>0 (dataset constant)
    # evaluation only (IS_REAL_WORLD runs, reference scripts/train_bunny_real.sh:54): the sensor's
    # ground-truth depth and the object mask behind results.gt_depth_mse / gt_object_depth_mse
    # (experiment_utils/get_results.py:47-51)
    gt_depth: Optional[torch.Tensor] = None     # [H,W], 0 = no measurement
    object_mask: Optional[torch.Tensor] = None  # [H,W] bool

    def valid_count(self) -> int:
        if self.n_valid_depth is None:
            self.n_valid_depth = int((self.depth > 0).sum().item()) if self.depth is not None else 0
        return self.n_valid_depth

    def downscaled(self, d: int) -> "View":
        """This view at 1/d of its resolution (cached): the colour image resized bilinearly to (H // d, W // d) --
        what Splatfacto's ``_downscale_if_required`` does (torchvision resize of a tensor, no antialiasing) --
        depth and uncertainty by nearest neighbour, so that 0 stays "unsupervised" and no depth is invented across
        an object boundary; camera as ``Camera.downscaled``."""
        if d <= 1:
            return self
        cache = self.__dict__.setdefault("_downscaled", {})
        if d not in cache:
            H, W = self.rgb.shape[0] // d, self.rgb.shape[1] // d
            F = torch.nn.functional
            rgb = F.interpolate(self.rgb.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False,
                                antialias=False)[0].permute(1, 2, 0).contiguous()
            near = lambda t: None if t is None else F.interpolate(t[None, None].float(), size=(H, W), mode="nearest")[0, 0].to(t.dtype).contiguous()
            cache[d] = View(cam=self.cam.downscaled(d), rgb=rgb, depth=near(self.depth), uncertainty=near(self.uncertainty),
                            gt_depth=near(self.gt_depth), object_mask=near(self.object_mask))
        return cache[d]


class DepthGaussianSplattingModel:
    """The trainable scene: flat SoA parameters + the render/loss/step methods."""
    list_hint = True          # speculative budget: learn a bound on the longest tile list, skip the unused sort launches
    LIST_HINT_AFTER = 16      # ... after this many settled frames since the last refinement / replay

    def __init__(self, config: ModelConfig, params: GaussianParams):
        self.config = config
        self.params = params
        self.step = 0
        self.populate_modules()

    # -- nerfstudio Model surface -------------------------------------------------------------
    def populate_modules(self):
        self.optimizer = FusedAdam(self.params, self.config.lrs())
        self.budget = ops.IntersectBudget()
        self._sync_budget = ops.IntersectBudget()   # render / eval path when `budget` is sync-free
        self.fuse_adam = True   # single-process steps use the fused K8+K9 kernel when it applies
        self.dp_factored_sh = True   # data-parallel steps exchange colour gradients, not SH rows
        # single-process steps that know the next view: its K1 runs inside this step's optimizer kernel
        self.front_prefetch = os.environ.get("TGS_FRONT_PREFETCH", "1") != "0"
        self._color_block = self._color_all = self._color_rows = None
        self.last = {}

    @property
    def num_points(self) -> int:
        return self.params.N

    def active_sh_degree(self, step: Optional[int] = None) -> int:
        c = self.config
        max_deg = int(round(math.sqrt(self.params.K))) - 1
        if c.sh_degree_interval <= 0:
            return min(c.sh_degree, max_deg)
        return min((self.step if step is None else step) // c.sh_degree_interval, c.sh_degree, max_deg)

    def get_outputs(self, cam: Camera, sh_degree: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Camera -> {rgb, depth, accumulation} (differentiable; autograd path)."""
        p = self.params
        deg = self.active_sh_degree() if sh_degree is None else sh_degree
        # the autograd / eval path has no replay logic: it always bins with a synchronous budget (a
        # sync-free training budget would turn an eval view that needs more pairs into background)
        budget = self.budget if self.budget.sync else self._sync_budget
        rgb, depth_acc, alpha, radii = ops.render(p.means, p.log_scales, p.quats, p.opac_logit, p.sh,
                                                  cam, deg, budget=budget)
        depth = depth_acc / torch.clamp(alpha, min=1e-10)
        return dict(rgb=rgb, depth=depth[..., None], accumulation=alpha[..., None],
                    depth_acc=depth_acc, alpha=alpha, radii=radii)

    def depth_loss(self, depth_acc, alpha, view: View) -> torch.Tensor:
        """mean over valid (D_gt > 0) of (D_hat - D_gt)^2 [ / (uw * U + eps) ]."""
        c = self.config
        if c.depth_loss_type not in DEPTH_LOSS_TYPES:
            raise ValueError(c.depth_loss_type)
        m = view.depth > 0
        cnt = m.sum()
        dhat = depth_acc / torch.clamp(alpha, min=1e-10)
        r2 = (dhat - view.depth) ** 2
        # a view without an uncertainty map is supervised with the SIMPLE form (same rule as loss_spec)
        if c.depth_loss_type == "DEPTH_UNCERTAINTY_WEIGHTED_LOSS" and view.uncertainty is not None:
            r2 = r2 / (c.uncertainty_weight * view.uncertainty + c.depth_eps)
        return torch.where(m, r2, torch.zeros_like(r2)).sum() / torch.clamp(cnt, min=1)

    def get_loss_dict(self, outputs, view: View) -> Dict[str, torch.Tensor]:
        c = self.config
        H, W = view.rgb.shape[:2]
        l1 = (outputs["rgb"] - view.rgb).abs().mean()
        loss = {"main_loss": (1 - c.ssim_lambda) * l1}
        if c.ssim_lambda > 0:
            loss["main_loss"] = loss["main_loss"] + c.ssim_lambda * (1 - _SSIM.apply(outputs["rgb"], view.rgb))
        if c.depth_loss_mult > 0 and view.depth is not None:
            loss["depth_loss"] = c.depth_loss_mult * self.depth_loss(outputs["depth_acc"], outputs["alpha"], view)
        return loss

    @torch.no_grad()
    def get_metrics_dict(self, outputs, view: View) -> Dict[str, torch.Tensor]:
        mse = ((outputs["rgb"] - view.rgb) ** 2).mean()
        m = {"psnr": -10.0 * torch.log10(mse), "gaussian_count": torch.tensor(self.num_points)}
        if view.depth is not None:
            valid = view.depth > 0
            d = outputs["depth"][..., 0]
            m["depth_mse"] = ((d - view.depth)[valid] ** 2).mean() if valid.any() else torch.tensor(0.0)
        return m

    @torch.no_grad()
    def get_image_metrics_and_images(self, outputs, view: View):
        """Eval metrics with the key names the reference aggregates
        (experiment_utils/get_results.py:35-52); lpips needs pretrained weights and is omitted."""
        mse = ((outputs["rgb"] - view.rgb) ** 2).mean()
        ssim_sum, _ = ops.ssim_fwd_bwd(outputs["rgb"], view.rgb, want_grad=False)
        H, W = view.rgb.shape[:2]
        metrics = {"psnr": float(-10.0 * torch.log10(mse)), "ssim": float(ssim_sum) / (3 * H * W)}
        if view.depth is not None:
            valid = view.depth > 0
            d = outputs["depth"][..., 0]
            metrics["depth_mse"] = float(((d - view.depth)[valid] ** 2).mean()) if valid.any() else 0.0
            metrics["supervised_depth_mse"] = metrics["depth_mse"]
        if view.gt_depth is not None:   # keys the reference's aggregator averages when present
            d = outputs["depth"][..., 0]
            valid = view.gt_depth > 0
            metrics["gt_depth_mse"] = float(((d - view.gt_depth)[valid] ** 2).mean()) if valid.any() else 0.0
            if view.object_mask is not None:
                obj = valid & view.object_mask
                metrics["gt_object_depth_mse"] = float(((d - view.gt_depth)[obj] ** 2).mean()) if obj.any() else 0.0
        images = {"img": torch.cat([view.rgb, outputs["rgb"]], dim=1),
                  "depth": outputs["depth"], "accumulation": outputs["accumulation"]}
        return metrics, images

    # -- fused train step ---------------------------------------------------------------------
    def loss_spec(self, view: View) -> dict:
        c = self.config
        H, W = view.rgb.shape[:2]
        spec = dict(gt_rgb=view.rgb, l1_weight=(1 - c.ssim_lambda) / (3 * H * W))
        if c.depth_loss_mult > 0 and view.depth is not None:
            if c.depth_loss_type not in DEPTH_LOSS_TYPES:
                raise ValueError(c.depth_loss_type)
            cnt = max(view.valid_count(), 1)
            spec.update(gt_depth=view.depth, depth_weight=c.depth_loss_mult / cnt,
                        uncertainty=view.uncertainty if c.depth_loss_type == "DEPTH_UNCERTAINTY_WEIGHTED_LOSS" else None,
                        uncertainty_weight=c.uncertainty_weight, eps=c.depth_eps)
        return spec

    def forward_backward(self, view: View, want_v_xy: bool = False, fuse_adam: bool = False,
                         color_block: Optional[torch.Tensor] = None, begin_step: bool = True,
                         colors=None, prefetch=None, next_front=None):
        """Forward + loss + backward of one view into ``params.grad`` (overwritten) -- or, with
        ``fuse_adam``, straight through the optimizer update (K8+K9 fused, ``params.grad`` untouched).
        No host sync unless ``budget.sync``.  Returns device tensors (l1+depth tile losses, ssim sum)."""
        p, c, cam = self.params, self.config, view.cam
        deg = self.active_sh_degree()
        H, W = cam.H, cam.W
        splats, radii, group_base, tile_start, sorted_gid, status = ops.project_bin_sort(
            cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, self.budget, want_radii=want_v_xy,
            colors=colors, next_front=next_front)
        guard = None if self.budget.sync else status   # overflowed frame => optimizer kernels are no-ops
        rgb, depth_acc, fT, fidx = ops.rasterize_fwd(cam, splats, sorted_gid, tile_start)
        v_img, ssim_sum = None, None
        if (c.ssim_lambda > 0 and c.pipeline_ssim and cam.num_tiles >= c.pipeline_ssim_min_tiles
                and not torch.cuda.is_current_stream_capturing()):
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=rgb.device)
            partials, tile_loss, ssim_sum = ops.rasterize_bwd_ssim_pipelined(
                cam, splats, group_base, sorted_gid, tile_start, rgb, depth_acc, fT, view.rgb,
                -c.ssim_lambda / (3 * H * W), self.loss_spec(view), self._side_stream)
        else:
            if c.ssim_lambda > 0:
                ssim_sum, v_img = ops.ssim_fwd_bwd(rgb, view.rgb, weight=-c.ssim_lambda / (3 * H * W), reduce=False)
            partials, tile_loss = ops.rasterize_bwd(cam, splats, group_base, sorted_gid, tile_start, rgb,
                                                    depth_acc, fT, v_rgb=v_img,
                                                    loss=self.loss_spec(view), want_tile_loss=True)
        if fuse_adam:
            v_xy = self.optimizer.backward_and_step(cam, deg, splats, group_base, partials, want_v_xy,
                                                    begin=begin_step, guard=guard, prefetch=prefetch)
        elif color_block is not None:   # data-parallel: geometry gradients + colour-gradient block(s)
            if isinstance(color_block, (list, tuple)):
                # pipelined exchange: K8 is launched chunk by chunk by the caller (train_step), between the gathers
                rows, blocks = color_block
                v_xy = torch.empty(p.N, 2, dtype=torch.float32, device=p.flat.device) if want_v_xy else None
                out4 = p.grad_views()[:4]

                def backward_chunk(c, cam=cam, deg=deg, splats=splats, group_base=group_base, partials=partials,
                                   guard=guard, v_xy=v_xy):
                    ops.project_bwd_color(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, splats,
                                          group_base, partials, out4, blocks[c], guard=guard, rows=rows[c], v_xy=v_xy)
                self._backward_chunk = backward_chunk
            else:
                v_xy = ops.project_bwd_color(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, splats,
                                             group_base, partials, p.grad_views()[:4], color_block, want_v_xy,
                                             guard=guard)
        else:
            v_xy = ops.project_bwd(cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, splats,
                                   group_base, partials, out=p.grad_views(), want_v_xy=want_v_xy, guard=guard)[5]
        self.last = dict(rgb=rgb, depth_acc=depth_acc, final_T=fT, splats=splats, v_xy=v_xy, radii=radii,
                         tile_loss=tile_loss, ssim_sum=ssim_sum, status=status, guard=guard, view=view)
        return tile_loss, ssim_sum

    def loss_from(self, tile_loss, ssim_sum, view: View) -> Dict[str, torch.Tensor]:
        c = self.config
        H, W = view.rgb.shape[:2]
        t = tile_loss.sum(0)
        main = t[0]
        if ssim_sum is not None:   # per-block partial sums (or an already reduced scalar)
            main = main + c.ssim_lambda * (1 - ssim_sum.sum() / (3 * H * W))
        return {"main_loss": main, "depth_loss": t[1]}

    def spatial_sort(self) -> torch.Tensor:
        """Put the Gaussians (parameters, gradients, Adam moments, densification statistics) in 3-D
        Morton order (long-run Gaussians dealt over the groups once a frame has been rendered: ``balance_long_runs``).  Purely a memory-layout choice -- the scene and every result are the same up to
        the permutation -- that makes a binning group of 256 consecutive Gaussians project onto a few
        dozen tiles: K1 then counts per (group, tile) in LDS and issues one global atomic per touched
        tile instead of one per pair, and `k_fill_bins` writes runs instead of single 8-byte pairs.
        Call at start-up and after densification (the trainer and bench.py do).  Returns the
        permutation (new row i = old row perm[i])."""
        from .optim import balanced_order
        # the last frame's tile count per Gaussian (the packed rect of its record): Gaussians with LONG runs of tiles are
        # dealt evenly over the binning groups instead of sitting next to their spatial neighbours (optim.balanced_order)
        hits = None
        cams = list(getattr(self, "_recent_cams", {}).values()) if self.config.balance_long_runs else []
        if os.environ.get("TGS_BALANCE_LONG_RUNS", "1") == "0":     # A/B switch
            cams = []
        if cams:
            # tiles per Gaussian summed over the cameras of the last steps (K1 alone, once per camera: the sort runs every
            # few hundred steps): a Gaussian that is long in one view of a few-view orbit is short in another, and a
            # layout balanced for ONE view left groups of 22 000 pairs in the next (profiles/r6_b_ckpt_loop_bunny.json)
            p, hits = self.params, 0
            for cam in cams:
                sp = ops.project_fwd(cam, p.means, p.log_scales, p.quats, p.opac_logit, None, -1)
                rect = sp[:, 10].contiguous().view(torch.int32)
                hits = hits + ((rect >> 16) & 255) * ((rect >> 24) & 255)
            dp = getattr(self, "_dp", None)
            if dp is not None and dp.active:
                # data parallel: every rank has seen ITS views only, and the replicas must agree on the row order -- sum the
                # counts over the ranks (the same collective on every rank: re-sorts happen at refinements, which are
                # barrier points) and divide by the rank-independent total
                import torch.distributed as dist
                n_cams = torch.tensor([len(cams)], dtype=torch.int64, device=hits.device)
                hits = hits.to(torch.int64)
                dist.all_reduce(hits, op=dist.ReduceOp.SUM)
                dist.all_reduce(n_cams, op=dist.ReduceOp.SUM)
                hits = (hits + int(n_cams) - 1) // int(n_cams)
            else:
                hits = (hits + len(cams) - 1) // len(cams)    # mean tiles per view (rounded up)
        long_run = ops.set_long_run()
        if hits is not None:
            # which Gaussians count as long runs (tgs_set_long_run: the kernels' counting box, K8's shared sums AND the rows
            # dealt over the groups here -- one number): 32 tiles, but 8 where the long runs ARE the frame -- an object on a
            # table: 7 % of the Gaussians hold 60 % of the pairs, and a group's pairs set the length of K1's counting,
            # k_fill_bins and K8 (largest group 8 951 -> 5 098 pairs, step -2.9 % / -2.2 % on the saved 720p checkpoints;
            # cfg3 -0.5 ... -1.3 % if it were applied there: profiles/r6_ab_runs.txt).  ModelConfig.long_run fixes it.
            long_run = self.config.long_run
            if long_run <= 0:
                share = float(hits[hits > 32].sum()) / max(float(hits.sum()), 1.0)
                long_run = 8 if share > 0.3 else 32
            if long_run != ops.set_long_run():
                ops.set_long_run(long_run)
            if self.config.long_run <= 0:
                # the same regime wants K6 to split more of its lists (tgs_set_k6_split_shape: -1.7 % of the step on the 720p
                # checkpoints, +1 % on uniform frames); the process's own settings are kept for everything else
                if getattr(self, "_k6_rule", None) is None:
                    self._k6_rule = (ops.set_k6_split(), ops.set_k6_split_shape())
                if share > 0.3:
                    ops.set_k6_split(1); ops.set_k6_split_shape(128, 2048)
                else:
                    ops.set_k6_split(self._k6_rule[0]); ops.set_k6_split_shape(*self._k6_rule[1])
        perm = balanced_order(self.params.means, hits, long_run=long_run)
        self.params.permute_(perm, self.optimizer.exp_avg, self.optimizer.exp_avg_sq)
        density = getattr(self, "density", None)
        if density is not None:
            density.grad_norm_sum = density.grad_norm_sum[perm]
            density.vis_count = density.vis_count[perm]
            density.max_radius = density.max_radius[perm]
        self._graphs = {}
        self._prefetch_ready = None
        self._refines_since_sort = 0
        return perm

    def enable_densification(self, cfg=None):
        """Turn on Splatfacto-style clone / split / cull refinement (touch_gs_amd.densify)."""
        from .densify import DensifyConfig, DensityController
        self.density = DensityController(cfg or DensifyConfig(), self.params.N, self.params.flat.device)

    # -- sync-free intersection budget ---------------------------------------------------------
    def enable_speculative_budget(self, capacity: int = 0, max_in_flight: int = 4) -> None:
        """Train without the per-step read-back of the intersection count.  Densification: every
        refinement is a barrier -- the pending verdicts are settled (and overflowed steps replayed) before
        the Gaussians change, overflowed frames add nothing to the refinement statistics, and the
        capacity follows the Gaussian count afterwards.  Every
        step's status word is copied to pinned host memory asynchronously and looked at a few steps
        later.  Data parallel (factored exchange only): the ranks agree on the overflow verdict on the
        device (the flag rides in the pad of the all-gathered colour block, ops.dp_agree_overflow), the
        optimizer kernels of ALL ranks are no-ops from the overflowing step on, and every rank inspects
        the verdict of step s - max_in_flight exactly at step s -- so all ranks notice at the same
        step and replay the same steps (the collectives stay matched).  Correctness does not depend on guessing the capacity right: a
        frame that overflows sets a sticky device word that empties every later frame and turns the
        guarded optimizer kernels into no-ops, so when the host notices it clears the word, grows the
        buffers, rewinds its step counters and replays the affected views -- the model ends up exactly
        where the synchronous budget would have put it (tests/test_gpu_api_surfaces.py)."""
        self.budget = ops.IntersectBudget(capacity=capacity, sync=False, speculative=True)
        self._pending = collections.deque()
        self._max_in_flight = max_in_flight
        self._pinned = []
        self._dp_status = None      # device {0, any rank overflowed} of the last data-parallel step
        self._dp = None
        # longest tile list the settled frames have shown (status[3]) -> the budget's max_list_hint: after LIST_HINT_AFTER
        # good frames the sort launches for list classes beyond 1.5 x that are no longer issued (two near-empty launches,
        # ~9 us per step at cfg3); a frame that breaks the bound is void and replayed like an overflow, with the hint off
        self._seen_longest, self._good_frames = 0, 0

    def _speculative_track(self, view: View, distributed: bool = False) -> None:
        # pinned int32[7]: {#intersections, overflow, sufficient capacity, -} of this rank's frame | the verdict | the
        # peer transport's error word (a wait that timed out; the device has already voided the step, see tgs_peer_wait)
        host = self._pinned.pop() if self._pinned else torch.zeros(7, dtype=torch.int32).pin_memory()
        host[:4].copy_(self.budget.last_status4, non_blocking=True)
        # data parallel: the verdict is the agreed flag, not this rank's own.  Single process: the frame's own status word
        # IS words 0, 1 of the four just copied (one 4.6 us copy on the stream per step instead of two, round 6)
        if distributed:
            host[4:6].copy_(self._dp_status, non_blocking=True)
        peer = getattr(self._dp, "peer", None) if distributed else None
        if peer is not None:
            host[6:].copy_(peer.err_word, non_blocking=True)
        else:
            host[6] = 0
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((view, host, ev))
        if distributed:   # deterministic: every rank looks at step s - max_in_flight exactly at step s
            if len(self._pending) > self._max_in_flight:
                self._speculative_poll(block=True, only_one=True)
        else:
            self._speculative_poll(block=len(self._pending) > self._max_in_flight)

    def _speculative_poll(self, block: bool = False, drain: bool = False, only_one: bool = False) -> None:
        while self._pending and (drain or block or (not only_one and self._dp is None and self._pending[0][2].query())):
            view, host, ev = self._pending[0]
            ev.synchronize()
            block = False
            if int(host[6]) != 0:      # surfaces at most max_in_flight steps after the wait gave up, before any replay
                self._dp.peer.raise_if(int(host[6]))
            if int(host[5] if self._dp is not None else host[1]) == 0:
                self._pending.popleft()
                self._seen_need = max(getattr(self, "_seen_need", 0), int(host[2]), int(host[0]))
                self._seen_longest = max(self._seen_longest, int(host[3]))
                self._good_frames += 1
                # armed once every training view of the current resolution level can have been seen (with more
                # than 16 views the bound came from a subset, and a later view beyond 1.5 x + 64 voided its frame)
                dens = getattr(self, "density", None)
                need_frames = max(self.LIST_HINT_AFTER, dens.cfg.num_train_data if dens is not None else 0,
                                  len(getattr(self, "_recent_cams", ())))
                if self.list_hint and self._good_frames >= need_frames:
                    self.budget.max_list_hint = max(self.budget.max_list_hint, int(1.5 * self._seen_longest) + 64)
                self._pinned.append(host)
                if only_one:
                    return
                continue
            # overflow: this step and everything enqueued after it did nothing on the device
            torch.cuda.synchronize()
            redo = list(self._pending)
            self._pending.clear()
            need = max(int(h[0]) for _, h, _ in redo)      # this rank's own largest frame
            self.budget.capacity = max(self.budget.capacity, int(need * self.budget.growth) + 1024)
            # (if a list longer than the hint voided the frame: every sort class again until enough frames have been seen)
            self._seen_longest = max([self._seen_longest] + [int(h[3]) for _, h, _ in redo])
            self.budget.max_list_hint, self._good_frames = -1, 0
            self.budget.sticky.zero_()
            self._prefetch_ready = None     # announced by a voided step: its front buffers were cleared under the sticky word
            self.optimizer.t -= len(redo)
            self.step -= len(redo)
            self.speculative_replays = getattr(self, "speculative_replays", 0) + len(redo)
            for v, h, _ in redo:
                self._pinned.append(h)
                self.train_step(v, self._dp)

    def flush(self) -> None:
        """Wait for every enqueued step and settle pending overflow checks (call before reading
        ``last``, saving a checkpoint or evaluating when the speculative budget is on)."""
        if getattr(self, "_pending", None) is not None and self.budget.speculative:
            self._speculative_poll(drain=True)

    # -- hipGraph replay of the step (small scenes are launch bound) ----------------------------
    def capture_step_graphs(self, views, headroom: float = 1.3, share_pool: bool = True) -> None:
        """Capture the fused single-process train step of every view into a hipGraph (via
        torch.cuda.CUDAGraph: memset + K1..K8/K9 nodes, one graph per view because the camera and
        the ground-truth pointers are launch arguments; all graphs share one memory pool).
        ``train_step`` then replays the view's graph: one launch per step instead of ~12, no Python
        between the kernels.  The only per-step scalars, Adam's bias corrections, live in device
        memory (TgsAdamSpec.device_bias_corr).  The intersection buffers are sized once from the
        views (x ``headroom``) and the overflow flag is checked lazily (``budget.check``).
        Graphs are dropped when the active SH degree or the number of Gaussians changes."""
        opt, deg = self.optimizer, self.active_sh_degree()
        if not opt.can_fuse_with_backward(deg):
            raise RuntimeError("step graphs need the fused K8+K9 path (dense SH at its full degree 1 or 3)")
        p = self.params
        need = 0
        for v in views:
            v.valid_count()
            b = ops.IntersectBudget()
            ops.project_bin_sort(v.cam, p.means, p.log_scales, p.quats, p.opac_logit, p.sh, deg, b)
            need = max(need, b.last_need)   # capacity under the per-XCD split, not the plain pair count
        self.budget = ops.IntersectBudget(capacity=int(need * headroom) + 4096, sync=False)
        self.budget.sticky_word(p.flat.device)   # allocated eagerly, outside the captured region
        opt.use_device_bias_corr = True
        opt.upload_bias_corr()
        torch.cuda.synchronize()
        self._graphs, pool = {}, None
        for v in views:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                self.forward_backward(v, fuse_adam=True, begin_step=False)
            pool = (pool or g.pool()) if share_pool else None
            self._graphs[id(v)] = (g, self.last, deg, p.N)
        opt.use_device_bias_corr = False   # only the captured launches read the device scalars

    def drop_step_graphs(self) -> None:
        self._graphs = {}
        self.optimizer.use_device_bias_corr = False

    def train_step(self, view: View, dp=None, next_view: Optional[View] = None) -> None:
        """One optimizer iteration on one view (per rank).  ``dp``: a parallel.GradSync or None.
        ``next_view``: the view of the FOLLOWING train_step call, if the caller knows it (single
        process, fused optimizer): the optimizer kernel then also evaluates the colours the updated
        Gaussians show to that camera, and the following step's K1 skips the SH rows (colour
        prefetch; same results bit for bit).  The caller promises not to modify the parameters
        between the two calls except through this class."""
        distributed = dp is not None and dp.active
        opt = self.optimizer
        if self.config.balance_long_runs and self.config.spatial_sort:
            # the cameras of the last steps (full resolution), for the next spatial_sort (host-side bookkeeping only)
            rc = self.__dict__.setdefault("_recent_cams", collections.OrderedDict())
            rc[id(view.cam)] = view.cam
            rc.move_to_end(id(view.cam))
            if len(rc) > 16:
                rc.popitem(last=False)
        opt.lrs["means"] = self.config.lr_means_at(self.step)   # scheduled position learning rate
        full_view = view    # what the speculative budget replays (the schedule is applied again on the replay)
        if self.config.num_downscales > 0:   # coarse-to-fine: this step (and the announced next one) at their resolutions
            d = self.config.downscale_factor(self.step)
            if d != getattr(self, "_last_downscale", d) and getattr(self, "_pending", None) is not None:
                # a new resolution level: the lists get ~4x longer -- learn the list bound again instead of voiding a frame
                self._seen_longest, self._good_frames = 0, 0
                self.budget.max_list_hint = -1
            self._last_downscale = d
            view = view.downscaled(d)
            if next_view is not None:
                next_view = next_view.downscaled(self.config.downscale_factor(self.step + 1))
        deg = self.active_sh_degree()
        pre, self._prefetch_ready = getattr(self, "_prefetch_ready", None), None
        graphs = getattr(self, "_graphs", None)
        if graphs and not distributed and getattr(self, "density", None) is None:
            entry = graphs.get(id(view))
            if entry is not None and entry[2] == deg and entry[3] == self.params.N:
                opt.begin_step()
                opt.upload_bias_corr()
                entry[0].replay()
                self.last = entry[1]
                self.step += 1
                return
            if entry is not None:
                self.drop_step_graphs()
        fuse = (not distributed) and self.fuse_adam and opt.can_fuse_with_backward(deg)
        factored = distributed and self.dp_factored_sh and opt.can_gather_sh()
        density = getattr(self, "density", None)
        if self.budget.speculative and distributed and not factored:
            raise RuntimeError("the speculative intersection budget does not combine with the dense (flat all-reduce) "
                               "data-parallel exchange")
        self._dp = dp if distributed else None
        block = None
        if factored:
            # the rank's colour-gradient blocks, one per row chunk of the pipelined exchange: [3 rows + 4] each
            # (colour gradients | camera position | overflow flag), and their all-gathered images [world, 3 rows + 4]
            rows = dp.color_chunk_rows(self.params.N)
            n = 3 * self.params.N + 4 * len(rows)
            if (self._color_block is None or self._color_block.numel() != n or self._color_all.numel() != n * dp.world
                    or self._color_rows != rows):
                dev = self.params.flat.device
                self._color_block = torch.zeros(n, dtype=torch.float32, device=dev)
                self._color_all = torch.zeros(dp.world * n, dtype=torch.float32, device=dev)
                self._color_rows, self._color_blocks, self._color_blocks_all, off = rows, [], [], 0
                for b, e in rows:
                    m = 3 * (e - b) + 4
                    self._color_blocks.append(self._color_block[off:off + m])
                    self._color_blocks_all.append(self._color_all[dp.world * off:dp.world * (off + m)].view(dp.world, m))
                    off += m
            block = (rows, self._color_blocks)
        # prefetched by the previous step: colours (+ front) from the fused optimizer kernel, or the front alone from
        # the geometry Adam of a data-parallel step
        colors = pre if (pre is not None and (fuse or (factored and pre.front_issued and not pre.colors_valid))
                         and pre.matches(view.cam, self.params.N, deg)) else None
        arm = None
        if fuse and next_view is not None and self.active_sh_degree(self.step + 1) == deg:
            bufs = getattr(self, "_prefetch_bufs", None)
            N, dev = self.params.N, self.params.flat.device
            if bufs is None or bufs[0].N != N or bufs[0].colors.device != dev:
                bufs = self._prefetch_bufs = [ops.ColorPrefetch(N, dev), ops.ColorPrefetch(N, dev)]
            # two buffers alternate: the one this step's K1 reads is not the one this step's K9 writes
            # front prefetch: the optimizer kernel also runs the next view's K1 into that frame's buffers
            front = None
            if self.front_prefetch and not torch.cuda.is_current_stream_capturing():
                front = ops.FrontBuffers(next_view.cam, N, self.budget.initial(N), density is not None, dev)
            arm = (bufs[1] if colors is bufs[0] else bufs[0]).arm(next_view.cam, deg, front, self.budget)
        if (factored and next_view is not None and self.front_prefetch
                and self.active_sh_degree(self.step + 1) == deg):
            # data-parallel form of the front prefetch: the geometry Adam (last kernel of the step) also runs this
            # rank's next K1; no colours (the SH rows are stepped by another kernel)
            N, dev = self.params.N, self.params.flat.device
            bufs = getattr(self, "_prefetch_bufs", None)
            if bufs is None or bufs[0].N != N or bufs[0].colors.device != dev:
                bufs = self._prefetch_bufs = [ops.ColorPrefetch(N, dev), ops.ColorPrefetch(N, dev)]
            front = ops.FrontBuffers(next_view.cam, N, self.budget.initial(N), density is not None, dev)
            arm = (bufs[1] if pre is bufs[0] else bufs[0]).arm(next_view.cam, deg, front, self.budget, colors_valid=False)
        self.forward_backward(view, want_v_xy=density is not None, fuse_adam=fuse, color_block=block,
                              colors=colors, prefetch=arm if fuse else None,
                              next_front=arm.front if arm is not None else None)
        self._prefetch_ready = arm
        if density is not None and not factored:
            # a frame that overflowed its intersection buffer (sync-free budget) rendered nothing and will be
            # replayed: it must not count as a view (guard = its status word, evaluated on the device)
            density.accumulate(self.last["v_xy"], self.last["radii"], view.cam.W, view.cam.H, guard=self.last["guard"],
                               view_key=id(full_view))
        if factored:
            dguard = None
            if self.budget.speculative:   # agree the overflow verdict across ranks before anything touches the model
                if self._dp_status is None:
                    self._dp_status = torch.zeros(2, dtype=torch.int32, device=self.params.flat.device)
                # a peer-exchange wait that times out raises both words: this step and every later one are voided.  Set every
                # step: a refinement replaces the budget, and with it the sticky word the transport would otherwise keep pointing at
                dp.set_poison_words(self.budget.sticky_word(self.params.flat.device), self._dp_status)
                dguard = self._dp_status

            rows = self._color_rows

            # the whole tail (SH Adam of every chunk + geometry Adam + the next view's K1) as ONE launch behind the
            # geometry all-reduce, where that is a gain (GradSync.fused_tail: by world size -- on a node the chunked SH
            # Adam hides under the transfers, which the fused launch cannot)
            fused_tail = (arm is not None and not fuse and len(rows) <= 8 and opt.p.K >= 4 and self.params.N > 0 and dp.fused_tail())

            def step_sh_chunk(c, allc, scale):
                if c == 0 and dguard is not None:   # every chunk block of a rank carries the frame's flag
                    ops.dp_agree_overflow(dp.world, rows[0][1] - rows[0][0], allc, dguard, self.budget.sticky)
                if not fused_tail:
                    opt.step_sh_gathered(dp.world, deg, allc, scale, guard=dguard, rows=rows[c])

            if fused_tail:
                step_geom = lambda b, e, scale: opt.step_sh_gathered_geom_and_project_next(
                    dp.world, deg, rows, self._color_blocks_all, scale, dguard, arm)
            elif arm is not None and not fuse:
                step_geom = lambda b, e, scale: opt.step_geom_and_project_next(deg, scale, dguard, arm)
            else:
                step_geom = lambda b, e, scale: opt.step_range(b, e, scale, guard=dguard)
            dp.pipelined_color_exchange_and_step(
                self.params.grad[:opt.geom_end()], self._color_blocks, self._color_blocks_all, self._backward_chunk,
                step_sh_chunk, step_geom, opt.begin_step)
            self._backward_chunk = None
            if density is not None:   # v_xy exists once the K8 chunks are enqueued
                # guarded by the AGREED verdict (written by chunk 0's dp_agree_overflow, enqueued above), not by this
                # rank's own status word: if another rank's frame overflowed, the step is voided on every rank and
                # replayed, and this rank's (valid) frame must not be counted twice
                # (data parallel: every rank sees its own views; the statistics are summed over the ranks at the
                # refinement, the identities are not -- the window counts as complete by its length, as for anonymous views)
                density.accumulate(self.last["v_xy"], self.last["radii"], view.cam.W, view.cam.H,
                                   guard=dguard if dguard is not None else self.last["guard"])
        elif distributed:
            dp.reduce_and_step(self.params.grad, self.optimizer.step_range, self.optimizer.begin_step)
        elif not fuse:
            self.optimizer.step(guard=self.last["guard"])
        self.step += 1
        if self.budget.speculative:
            self._speculative_track(full_view, distributed)
        elif self.budget.sync and distributed and getattr(dp, "peer", None) is not None:
            # synchronous budget: no guard words exist, so the error word is read every step (this mode syncs per step
            # anyway).  A pre-sized sync-free budget (bench.py) neither syncs nor polls: its caller checks the transport
            # at its own sync point (GradSync.check_transport), as it checks the budget.
            dp.peer.raise_if(int(dp.peer.err_word.item()))
        if density is not None and density.due(self.step) and getattr(self, "_refined_at", None) != self.step:
            if self.budget.speculative:
                # barrier: settle every pending overflow verdict first (a drain that finds one replays the
                # affected steps -- possibly including this one, whose replay then refines; `_refined_at`
                # keeps the outer call from refining a second time)
                self.flush()
                if getattr(self, "_refined_at", None) == self.step:
                    return
            n_before = self.params.N
            self.params, self.optimizer, self.last_refine = density.refine(self.params, self.optimizer, self.step, dp)
            self._refined_at = self.step
            if self.budget.speculative:
                # the intersection count follows N: size the new buffers from the largest capacity any frame since
                # the last refinement needed (the verdicts just drained carry it), scaled by the growth of N, +25 %;
                # a wrong guess only costs a replay.  (Never from the old capacity: that would compound.)
                seen = getattr(self, "_seen_need", 0)
                ratio = max(1.0, self.params.N / max(n_before, 1))
                cap = int(seen * ratio * 1.25) + 4096 if seen > 0 else int(self.budget.capacity * ratio) + 4096
                self._seen_need = 0
                self._seen_longest, self._good_frames = 0, 0     # the lists change with the Gaussians: learn the bound again
                self.budget = ops.IntersectBudget(capacity=cap, sync=False, speculative=True)
            else:
                self.budget = ops.IntersectBudget()  # the intersection count changes with N
            self._prefetch_ready = None
            self._graphs = {}
            if self.config.spatial_sort:
                # refine() puts every spawned row directly behind its parent, so the buffer stays spatially
                # ordered; the full re-sort (argsort + gathers of parameters and both moment buffers) only
                # corrects the slow drift of the means, every `resort_every_refines` refinements
                self._refines_since_sort = getattr(self, "_refines_since_sort", 0) + 1
                if self._refines_since_sort >= self.config.resort_every_refines:
                    self.spatial_sort()

    # -- checkpoint -----------------------------------------------------------------------------
    def state_dict(self):
        return dict(flat=self.params.flat, N=self.params.N, K=self.params.K, step=self.step,
                    optim=self.optimizer.state_dict(), config=dataclasses.asdict(self.config))

    def load_state_dict(self, sd):
        """Restores parameters, Adam state and the step counter.  A checkpoint written after
        densification holds a different number of Gaussians: the flat stores are then re-allocated
        at the checkpoint's size (same SH storage degree required)."""
        if sd["K"] != self.params.K:
            raise ValueError(f"checkpoint stores {sd['K']} SH bases per Gaussian, the model {self.params.K}")
        if sd["N"] != self.params.N:
            dev = self.params.flat.device
            self.params = GaussianParams.allocate(sd["N"], sd["K"], dev)
            self.optimizer = FusedAdam(self.params, self.config.lrs())
            self.budget = ops.IntersectBudget()
            self._color_block = self._color_all = None
            self._graphs = {}
            if getattr(self, "density", None) is not None:
                self.density.reset_stats(sd["N"], dev)
        self.params.flat.copy_(sd["flat"])
        self._prefetch_ready = None
        self.step = sd["step"]
        self.optimizer.load_state_dict(sd["optim"])


class _SSIM(torch.autograd.Function):
    """Mean SSIM through the K10 kernel, differentiable w.r.t. the rendered image."""

    @staticmethod
    def forward(ctx, img, gt):
        H, W = img.shape[:2]
        tot, v = ops.ssim_fwd_bwd(img.contiguous(), gt.contiguous(), weight=1.0 / (3 * H * W))
        ctx.save_for_backward(v)
        return tot / (3 * H * W)

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        return v * g, None
