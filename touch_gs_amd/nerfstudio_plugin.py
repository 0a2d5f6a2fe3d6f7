"""nerfstudio method plugin ``depth-gaussian-splatting`` (import-guarded).

The reference trains through a nerfstudio fork's method of this name
(scripts/train_bunny_real.sh:52); the only in-tree evidence of how Touch-GS plugs into nerfstudio
is the legacy NeRF plugin (legacy/config_tactile.py:23-56: MethodSpecification(TrainerConfig(
method_name, pipeline=VanillaPipelineConfig(datamanager, model), optimizers), description)), which
this file mirrors.  nerfstudio is not installable here (no network), so the MethodSpecification and
the shell classes are only built when it imports and have only executed against a stand-in for the
nerfstudio API (tests/test_gpu_api_surfaces.py::test_nerfstudio_plugin_*); everything the shell
delegates to is free of nerfstudio imports and IS exercised by the GPU tests:

* ``AutogradGaussians`` -- parameter groups, differentiable render, loss with nerfstudio batch shapes,
  SH-degree ramp;
* ``supervision_table`` / ``load_supervision_maps`` -- the tactile data path: ``depth_file_path`` and
  ``uncertainty_file_path`` of every frame (written by the reference's
  utils/add_depth_file_path_to_transforms.py:37-50), uint16-mm PNGs -> scene units exactly as the
  in-tree dataparser treats depth (legacy/dataparser_tactile.py:159-162 file list, :65-66 the mm -> m
  factor, :301-312 metadata; the factor is multiplied by the dataparser scale in the dataset) ->
  ``batch["depth_image"]`` / ``batch["uncertainty"]`` [H,W,1];
* ``ParamGroupRefiner`` -- Splatfacto-style densify / cull / opacity reset on the six
  ``nn.Parameter`` groups including the surgery on their torch.optim.Adam states; driven under
  nerfstudio by ``get_training_callbacks``.

``touch_gs_amd.train`` is the self-contained trainer (fused path).

Register with  [project.entry-points."nerfstudio.method_configs"]
               depth-gaussian-splatting = "touch_gs_amd.nerfstudio_plugin:depth_gaussian_splatting"
"""
from __future__ import annotations

METHOD_NAME = "depth-gaussian-splatting"
DESCRIPTION = "Touch-GS: Gaussian splatting with tactile depth + uncertainty supervision (MI355X HIP rasterizer)"

try:  # pragma: no cover - nerfstudio is absent in the build container
    from nerfstudio.configs.base_config import ViewerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.plugins.types import MethodSpecification
    available = True
except Exception:  # noqa: BLE001
    available = False


def model_flags():
    """The tyro flags the reference passes (scripts/train_*.sh) and their fields on ModelConfig."""
    return {"--pipeline.model.depth-loss-mult": "depth_loss_mult",
            "--pipeline.model.depth-loss-type": "depth_loss_type",
            "--pipeline.model.uncertainty_weight": "uncertainty_weight"}


# Splatfacto's parameter-group names and Adam learning rates; "xyz" decays
# exponentially to XYZ_LR_FINAL over the run.
PARAM_GROUP_LRS = {"xyz": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.000125, "opacity": 0.05,
                   "scaling": 0.005, "rotation": 0.001}
XYZ_LR_FINAL = 1.6e-6


def supervision_table(data_dir: str, transforms_name: str = "transforms.json"):
    """{absolute image path: (depth path or None, uncertainty path or None)} from the frames of
    ``transforms.json`` (keys ``file_path`` / ``depth_file_path`` / ``uncertainty_file_path``,
    reference utils/add_depth_file_path_to_transforms.py:37-50).  Keyed by image path so that a
    dataparser's split / ordering (legacy/dataparser_tactile.py:199-240) carries over to both maps."""
    import json
    import os
    with open(os.path.join(data_dir, transforms_name)) as f:
        meta = json.load(f)
    root = os.path.abspath(data_dir)
    full = lambda rel: None if rel is None else os.path.normpath(os.path.join(root, rel))
    return {full(fr["file_path"]): (full(fr.get("depth_file_path")), full(fr.get("uncertainty_file_path")))
            for fr in meta["frames"] if "file_path" in fr}


def _resize_nearest(a, H: int, W: int):
    """Nearest-neighbour resize to the camera's resolution (what nerfstudio's depth reader does with
    cv2.INTER_NEAREST); identity when the sizes agree."""
    import numpy as np
    if a.shape[0] == H and a.shape[1] == W:
        return a
    yi = np.minimum((np.arange(H) * (a.shape[0] / H)).astype(np.int64), a.shape[0] - 1)
    xi = np.minimum((np.arange(W) * (a.shape[1] / W)).astype(np.int64), a.shape[1] - 1)
    return a[yi][:, xi]


def load_supervision_maps(depth_path, uncertainty_path, H: int, W: int, depth_unit_scale_factor: float = 1e-3,
                          dataparser_scale: float = 1.0, uncertainty_scaling: str = "linear",
                          uncertainty_floor: float = 0.0):
    """One frame's tactile supervision as float32 arrays [H,W,1] (None where the frame has no file).

    depth       = uint16 PNG value x depth_unit_scale_factor x dataparser_scale  (0 = unsupervised),
                  the nerfstudio DepthDataset convention the in-tree dataparser feeds
                  (legacy/dataparser_tactile.py:65-66,301-312; config_tactile.py:33 wires DepthDataset);
    uncertainty = the same decoding x dataset.uncertainty_factor(uncertainty_scaling, ..) -- see
                  touch_gs_amd/dataset.py on the (UNVERIFIED-PRIOR) choice of units; both trainers share it.
                  ``uncertainty_floor``: lower bound in the map's own units, applied before the scaling (dataset.py)."""
    import numpy as np
    from .dataset import uncertainty_factor
    from .plumbing import from_uint16_mm, read_png16
    out = []
    for path, factor in ((depth_path, depth_unit_scale_factor * 1e3 * dataparser_scale),
                         (uncertainty_path, uncertainty_factor(uncertainty_scaling, depth_unit_scale_factor, dataparser_scale))):
        if path is None:
            out.append(None)
            continue
        a = _resize_nearest(from_uint16_mm(read_png16(path)), H, W)
        if path is uncertainty_path and uncertainty_floor > 0:
            a = np.maximum(a, uncertainty_floor)
        out.append((a * factor).astype(np.float32)[..., None])
    return out[0], out[1]


def load_touch_seed_points(data_dir: str, transform=None, scale: float = 1.0):
    """The touch point cloud the reference's pipeline writes to seed the Gaussians -- ``points_touch.npy`` [M,3] and
    ``points_colors.npy`` [M,3] in 0..255 (utils/create_point_cloud_from_touches.py:243-244; colours x 255 at :171)
    -- moved into the dataparser's frame exactly as nerfstudio moves a COLMAP point cloud: homogeneous points times
    ``dataparser_transform`` (3x4), then times ``dataparser_scale``.  -> (xyz float32 [M,3], rgb float32 [M,3] in
    0..255) or None if the scene has no seed files.  (touch_gs_amd.dataset.Scene.seed_points is the same for the
    self-contained trainer.)"""
    import os
    import numpy as np
    p, c = os.path.join(data_dir, "points_touch.npy"), os.path.join(data_dir, "points_colors.npy")
    if not (os.path.exists(p) and os.path.exists(c)):
        return None
    xyz = np.load(p).astype(np.float64).reshape(-1, 3)
    rgb = np.load(c).astype(np.float32).reshape(-1, 3)
    if transform is not None:
        T = np.asarray(transform, dtype=np.float64).reshape(-1, 4)[:3]
        xyz = xyz @ T[:, :3].T + T[:, 3]
    return (xyz * float(scale)).astype(np.float32), rgb


class AutogradGaussians:
    """What a nerfstudio ``Model`` needs from this library, free of nerfstudio imports so that it is
    exercised by the GPU tests: the Gaussian parameters as six ``torch.nn.Parameter`` groups (the
    trainer's own optimizers step them through ``loss.backward()``), a differentiable render through
    the HIP kernels (``ops.render``), and the depth-supervised loss with nerfstudio batch shapes
    ([H,W,3] image, [H,W,1] depth_image / uncertainty)."""

    def __init__(self, config, means, colors01, device="cuda", init_scale: float = -4.0, init_opacity: float = -2.0,
                 seed: int = 0):
        import torch
        from .model import DepthGaussianSplattingModel, ModelConfig
        from .optim import GaussianParams
        N, K = means.shape[0], (config.sh_degree + 1) ** 2
        g = torch.Generator().manual_seed(seed)
        sh = torch.zeros(N, K, 3)
        sh[:, 0] = (colors01.cpu().float() - 0.5) / 0.28209479177387814
        quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
        P = torch.nn.Parameter
        dev = device
        self.params = {"xyz": P(means.float().to(dev).contiguous()),
                       "scaling": P(torch.full((N, 3), float(init_scale), device=dev)),
                       "rotation": P(quats.to(dev)),
                       "opacity": P(torch.full((N,), float(init_opacity), device=dev)),
                       "features_dc": P(sh[:, :1].to(dev).contiguous()),
                       "features_rest": P(sh[:, 1:].to(dev).contiguous())}
        mc = ModelConfig(sh_degree=config.sh_degree, ssim_lambda=config.ssim_lambda,
                         depth_loss_mult=config.depth_loss_mult, depth_loss_type=config.depth_loss_type,
                         uncertainty_weight=config.uncertainty_weight)
        # the core model supplies the loss / metric code; its own flat store stays empty (the
        # nerfstudio optimizers own the parameters)
        self.core = DepthGaussianSplattingModel(mc, GaussianParams.allocate(0, K, dev))
        self.sh_degree = config.sh_degree
        # Splatfacto's SH ramp: one more band every `sh_degree_interval` steps (0 = full degree at once)
        self.sh_degree_interval = int(getattr(config, "sh_degree_interval", 0) or 0)
        # Splatfacto's resolution schedule: render and supervise at 1 / 2^(num_downscales - step //
        # resolution_schedule) of the camera's resolution; evaluation always at full resolution
        self.num_downscales = int(getattr(config, "num_downscales", 0) or 0)
        self.resolution_schedule = int(getattr(config, "resolution_schedule", 250) or 0)
        self.training = True
        self.step = 0
        self.track_xy_grad = False      # set by ParamGroupRefiner: render() then keeps the screen-space gradient
        self.last_xy = self.last_radii = self.last_wh = None

    @property
    def num_points(self) -> int:
        return int(self.params["xyz"].shape[0])

    def param_groups(self):
        return {k: [v] for k, v in self.params.items()}

    def active_sh_degree(self) -> int:
        if self.sh_degree_interval <= 0:
            return self.sh_degree
        return min(self.step // self.sh_degree_interval, self.sh_degree)

    def downscale_factor(self) -> int:
        if not self.training or self.num_downscales <= 0 or self.resolution_schedule <= 0:
            return 1
        return 2 ** max(self.num_downscales - self.step // self.resolution_schedule, 0)

    def render(self, cam, sh_degree=None):
        import torch
        from . import ops
        p = self.params
        cam = cam.downscaled(self.downscale_factor())
        sh = torch.cat([p["features_dc"], p["features_rest"]], dim=1)
        xy = None
        if self.track_xy_grad and torch.is_grad_enabled():
            # INRIA `means2D.grad` convention: a zero [N,2] leaf that receives the screen-space mean gradient
            xy = torch.zeros(p["xyz"].shape[0], 2, device=p["xyz"].device, requires_grad=True)
        rgb, depth_acc, alpha, radii = ops.render(p["xyz"], p["scaling"], p["rotation"], p["opacity"], sh, cam,
                                                  self.active_sh_degree() if sh_degree is None else sh_degree,
                                                  means2d=xy, budget=self.core._sync_budget)
        if xy is not None:
            self.last_xy, self.last_radii, self.last_wh = xy, radii, (cam.W, cam.H)
        depth = depth_acc / torch.clamp(alpha, min=1e-10)
        return dict(rgb=rgb, depth=depth[..., None], accumulation=alpha[..., None], depth_acc=depth_acc, alpha=alpha,
                    radii=radii)

    @staticmethod
    def view_from_batch(batch, like):
        """nerfstudio batch -> View: depth_image / uncertainty arrive as [H,W,1]."""
        import torch
        from .model import View
        H, W = like.shape[0], like.shape[1]
        img = batch["image"].to(like).float()
        sq = lambda t: None if t is None else t.to(like.device).reshape(t.shape[0], t.shape[1]).float()
        d, u = sq(batch.get("depth_image")), sq(batch.get("uncertainty"))
        if img.shape[0] != H or img.shape[1] != W:   # the render is downscaled (resolution schedule): so is the supervision
            F = torch.nn.functional
            img = F.interpolate(img.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).contiguous()
            near = lambda t: None if t is None else F.interpolate(t[None, None], size=(H, W), mode="nearest")[0, 0].contiguous()
            d, u = near(d), near(u)
        return View(cam=None, rgb=img, depth=d, uncertainty=u)

    def loss_dict(self, outputs, batch):
        return self.core.get_loss_dict(outputs, self.view_from_batch(batch, outputs["rgb"]))

    def metrics_dict(self, outputs, batch):
        return self.core.get_metrics_dict(outputs, self.view_from_batch(batch, outputs["rgb"]))

    def image_metrics_and_images(self, outputs, batch):
        return self.core.get_image_metrics_and_images(outputs, self.view_from_batch(batch, outputs["rgb"]))


class ParamGroupRefiner:
    """Splatfacto-style refinement (densify.DensityController: clone / split / cull / opacity reset,
    SURVEY App. A.3) for Gaussians that live in six ``nn.Parameter`` groups stepped by torch.optim.Adam
    -- the situation under nerfstudio.  After every training iteration ``after_train_iteration`` adds
    the step's screen-space gradient norms, visibility and radii to the statistics; when a refinement is
    due the groups are packed into the flat store, refined there (the same code path as the
    self-contained trainer), unpacked into NEW Parameters, and every optimizer's state is rewritten for
    the new rows (surviving rows keep their moments, new rows start at zero, `step` is kept)."""

    GROUP_OF = dict(means="xyz", log_scales="scaling", quats="rotation", opac_logit="opacity")

    def __init__(self, gaussians: "AutogradGaussians", cfg=None):
        from .densify import DensifyConfig, DensityController
        self.g = gaussians
        self.cfg = cfg or DensifyConfig()
        self.ctrl = DensityController(self.cfg, gaussians.num_points, gaussians.params["xyz"].device)
        gaussians.track_xy_grad = True
        self.last_info = None
        self.on_replace = None      # callback(name, new_parameter): lets an nn.Module re-register the tensors

    def after_train_iteration(self, step: int, optimizers: dict) -> "dict | None":
        """``optimizers``: {group name: torch.optim.Optimizer} (nerfstudio: ``Optimizers.optimizers``).
        ``step`` counts finished iterations (1 after the first).  Returns the refine info when one ran."""
        g = self.g
        if g.last_xy is not None and g.last_xy.grad is not None:
            W, H = g.last_wh
            self.ctrl.accumulate(g.last_xy.grad, g.last_radii, W, H)
        g.last_xy = g.last_radii = None
        if not self.ctrl.due(step):
            return None
        self.last_info = self.refine(step, optimizers)
        return self.last_info

    @staticmethod
    def _adam_state(opt, p):
        import torch
        st = opt.state.get(p, {})
        z = lambda: torch.zeros_like(p.data)
        return st.get("exp_avg", z()), st.get("exp_avg_sq", z()), st.get("step", None)

    def refine(self, step: int, optimizers: dict) -> dict:
        import torch
        from .optim import FusedAdam, GaussianParams
        g, P = self.g, self.g.params
        with torch.no_grad():
            sh = torch.cat([P["features_dc"], P["features_rest"]], dim=1)
            flat = GaussianParams.from_tensors(P["xyz"].data, P["scaling"].data, P["rotation"].data, P["opacity"].data, sh)
            N, K = flat.N, flat.K
            opt = FusedAdam(flat, {})            # carries the moments through DensityController.refine
            mv, vv = GaussianParams.views_of(opt.exp_avg, N, K), GaussianParams.views_of(opt.exp_avg_sq, N, K)
            steps = {}
            for name, grp in self.GROUP_OF.items():
                m, v, steps[grp] = self._adam_state(optimizers[grp], P[grp])
                mv[name].copy_(m.view_as(mv[name])); vv[name].copy_(v.view_as(vv[name]))
            m0, v0, steps["features_dc"] = self._adam_state(optimizers["features_dc"], P["features_dc"])
            m1, v1, steps["features_rest"] = self._adam_state(optimizers["features_rest"], P["features_rest"])
            mv["sh"].copy_(torch.cat([m0, m1], dim=1)); vv["sh"].copy_(torch.cat([v0, v1], dim=1))
            new_flat, new_opt, info = self.ctrl.refine(flat, opt, step)
            n2 = new_flat.N
            mv2, vv2 = GaussianParams.views_of(new_opt.exp_avg, n2, K), GaussianParams.views_of(new_opt.exp_avg_sq, n2, K)
            new_vals = {"xyz": (new_flat.means, mv2["means"], vv2["means"]),
                        "scaling": (new_flat.log_scales, mv2["log_scales"], vv2["log_scales"]),
                        "rotation": (new_flat.quats, mv2["quats"], vv2["quats"]),
                        "opacity": (new_flat.opac_logit, mv2["opac_logit"], vv2["opac_logit"]),
                        "features_dc": (new_flat.sh[:, :1], mv2["sh"][:, :1], vv2["sh"][:, :1]),
                        "features_rest": (new_flat.sh[:, 1:], mv2["sh"][:, 1:], vv2["sh"][:, 1:])}
            for grp, (val, m, v) in new_vals.items():
                old = P[grp]
                newp = torch.nn.Parameter(val.clone().contiguous().view(-1, *old.shape[1:]) if old.dim() > 1
                                          else val.clone().contiguous())
                o = optimizers[grp]
                o.state.pop(old, None)
                st = {"exp_avg": m.clone().contiguous().view_as(newp), "exp_avg_sq": v.clone().contiguous().view_as(newp)}
                if steps[grp] is not None:
                    st["step"] = steps[grp]
                else:   # a capturable / fused torch Adam keeps `step` on the parameter's device
                    on_dev = any(pg.get("capturable") or pg.get("fused") for pg in o.param_groups)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=newp.device if on_dev else "cpu")
                o.state[newp] = st
                for pg in o.param_groups:
                    pg["params"] = [newp if q is old else q for q in pg["params"]]
                P[grp] = newp
                if self.on_replace is not None:
                    self.on_replace(grp, newp)
        return info


if available:  # pragma: no cover
    from dataclasses import dataclass, field
    from typing import Type

    from nerfstudio.cameras.cameras import Cameras
    from nerfstudio.data.datamanagers.full_images_datamanager import FullImageDatamanager, FullImageDatamanagerConfig
    from nerfstudio.data.dataparsers.nerfstudio_dataparser import Nerfstudio, NerfstudioDataParserConfig
    from nerfstudio.data.datasets.base_dataset import InputDataset
    from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackLocation
    from nerfstudio.engine.optimizers import AdamOptimizerConfig
    from nerfstudio.models.base_model import Model, ModelConfig as NSModelConfig
    from nerfstudio.pipelines.base_pipeline import VanillaPipelineConfig

    from .camera import Camera
    from .densify import DensifyConfig

    # ---- data: dataparser -> dataset -> datamanager ----------------------------------------------
    @dataclass
    class TactileDataParserConfig(NerfstudioDataParserConfig):
        """nerfstudio-data + the tactile maps.  The stock parser already lists ``depth_file_path`` as
        metadata["depth_filenames"] with ``depth_unit_scale_factor`` (the in-tree copy:
        legacy/dataparser_tactile.py:159-162,301-312); this one adds the per-frame
        ``uncertainty_file_path`` in the same (split-filtered) order."""
        _target: Type = field(default_factory=lambda: TactileDataParser)
        uncertainty_scaling: str = "linear"
        # dataset.py: a floor of 0.05 bounds the touch : vision weight ratio of the depth loss at 100 (the few-view preset of
        # touch_gs_amd.train); 0 = off = the reference's loss for the reference's flags (default)
        uncertainty_floor: float = 0.0

    class TactileDataParser(Nerfstudio):
        config: TactileDataParserConfig

        def _generate_dataparser_outputs(self, split="train"):
            out = super()._generate_dataparser_outputs(split)
            import os
            table = supervision_table(str(self.config.data))
            rows = [table.get(os.path.normpath(os.path.abspath(str(f))), (None, None)) for f in out.image_filenames]
            md = out.metadata
            if md.get("depth_filenames") is None and any(r[0] for r in rows):
                md["depth_filenames"] = [r[0] for r in rows]
            md["uncertainty_filenames"] = [r[1] for r in rows] if any(r[1] for r in rows) else None
            md.setdefault("depth_unit_scale_factor", self.config.depth_unit_scale_factor)
            md["uncertainty_scaling"] = self.config.uncertainty_scaling
            md["uncertainty_floor"] = self.config.uncertainty_floor
            # the touch point cloud is what seeds the model (the Touch-GS scenes have no COLMAP ply for
            # load_3D_points to find): nerfstudio hands metadata["points3D_xyz"/"points3D_rgb"] to the model as
            # `seed_points`
            if md.get("points3D_xyz") is None:
                import torch
                seeds = load_touch_seed_points(str(self.config.data), getattr(out, "dataparser_transform", None),
                                               getattr(out, "dataparser_scale", 1.0))
                if seeds is not None:
                    md["points3D_xyz"] = torch.from_numpy(seeds[0])
                    md["points3D_rgb"] = torch.from_numpy(seeds[1]).clamp(0, 255).to(torch.uint8)
            return out

    class TactileDepthDataset(InputDataset):
        """InputDataset + ``depth_image`` / ``uncertainty`` [H,W,1] per frame (what nerfstudio's
        DepthDataset does for depth, config_tactile.py:33, extended to the second map)."""
        # depth_image / uncertainty are NOT excluded from the datamanager's host-to-device move: they are read by
        # every loss / metrics call

        def __init__(self, dataparser_outputs, scale_factor: float = 1.0):
            super().__init__(dataparser_outputs, scale_factor)
            md = self.metadata
            self.depth_filenames = md.get("depth_filenames")
            self.uncertainty_filenames = md.get("uncertainty_filenames")
            self.depth_unit_scale_factor = md.get("depth_unit_scale_factor", 1e-3)
            self.uncertainty_scaling = md.get("uncertainty_scaling", "linear")
            self.uncertainty_floor = md.get("uncertainty_floor", 0.0)

        def get_metadata(self, data):
            import torch
            i = data["image_idx"]
            dpath = None if self.depth_filenames is None else self.depth_filenames[i]
            upath = None if self.uncertainty_filenames is None else self.uncertainty_filenames[i]
            if dpath is None and upath is None:
                return {}
            H, W = int(self._dataparser_outputs.cameras.height[i]), int(self._dataparser_outputs.cameras.width[i])
            d, u = load_supervision_maps(None if dpath is None else str(dpath), None if upath is None else str(upath), H, W,
                                         self.depth_unit_scale_factor, self._dataparser_outputs.dataparser_scale,
                                         self.uncertainty_scaling, self.uncertainty_floor)
            out = {}
            if d is not None:
                out["depth_image"] = torch.from_numpy(d)
            if u is not None:
                out["uncertainty"] = torch.from_numpy(u)
            return out

    # ---- model ------------------------------------------------------------------------------------
    @dataclass
    class DepthGSModelConfig(NSModelConfig):
        _target: Type = field(default_factory=lambda: DepthGSNerfstudioModel)
        depth_loss_mult: float = 0.2
        depth_loss_type: str = "DEPTH_UNCERTAINTY_WEIGHTED_LOSS"
        uncertainty_weight: float = 1.0
        sh_degree: int = 3
        sh_degree_interval: int = 1000
        ssim_lambda: float = 0.2
        num_random: int = 50000            # Gaussians drawn uniformly from the scene cube when there are no seed points ...
        # ... and, WITH a touch seed cloud, the random fill added next to it: the cloud only covers the touched object,
        # table and background have to come from somewhere (train.init_params does the same).  0 = seeds only
        random_fill: int = 50000
        max_seed_points: int = 50000       # a larger cloud (every touch-depth pixel of every training view) is subsampled; 0 = keep all
        num_downscales: int = 2            # Splatfacto's coarse-to-fine schedule
        resolution_schedule: int = 250
        # refinement (Splatfacto defaults)
        refine: bool = True
        warmup_length: int = 500
        refine_every: int = 100
        densify_grad_thresh: float = 0.0002
        densify_size_thresh: float = 0.01
        cull_alpha_thresh: float = 0.1
        cull_scale_thresh: float = 0.5
        reset_alpha_every: int = 30
        stop_split_at: int = 15000
        # densify.py: cull Gaussians unseen for a whole refinement window.  nerfstudio's datamanager draws the views at
        # random per epoch, so the window only counts as complete from refine_every >= 2 num_train_data - 1
        cull_unseen: bool = False

    class DepthGSNerfstudioModel(Model):
        """Thin shell: nerfstudio Model API -> AutogradGaussians / ParamGroupRefiner (above).  Parameters
        are nn.Parameters in Splatfacto's six groups, so nerfstudio's loss.backward() + optimizer steps
        train them, and the training callbacks densify / cull them."""
        config: DepthGSModelConfig

        def populate_modules(self):
            import torch
            seed = self.kwargs.get("seed_points")
            if seed is not None:
                means, cols = seed[0].float(), seed[1].float() / 255.0
                g = torch.Generator().manual_seed(0)
                if self.config.max_seed_points > 0 and len(means) > self.config.max_seed_points:
                    sel = torch.randperm(len(means), generator=g)[:self.config.max_seed_points]
                    means, cols = means[sel], cols[sel]
                if self.config.random_fill > 0:
                    means = torch.cat([means, ((torch.rand(self.config.random_fill, 3, generator=g) - 0.5) * 2).to(means)])
                    cols = torch.cat([cols, torch.rand(self.config.random_fill, 3, generator=g).to(cols)])
            else:
                means = (torch.rand(self.config.num_random, 3) - 0.5) * 2
                cols = torch.rand(self.config.num_random, 3)
            self.gaussians = AutogradGaussians(self.config, means, cols)
            self.gauss_params = torch.nn.ParameterDict(self.gaussians.params)   # registered: state_dict, .to()
            self.refiner = None
            if self.config.refine:
                c = self.config
                self.refiner = ParamGroupRefiner(self.gaussians, DensifyConfig(
                    warmup_length=c.warmup_length, refine_every=c.refine_every, densify_grad_thresh=c.densify_grad_thresh,
                    densify_size_thresh=c.densify_size_thresh, cull_alpha_thresh=c.cull_alpha_thresh,
                    cull_scale_thresh=c.cull_scale_thresh, reset_alpha_every=c.reset_alpha_every,
                    stop_split_at=c.stop_split_at, num_train_data=int(getattr(self, "num_train_data", 0) or 0),
                    cull_unseen=c.cull_unseen))
                self.refiner.on_replace = lambda name, p: self.gauss_params.__setitem__(name, p)

        def get_param_groups(self):
            return self.gaussians.param_groups()

        def get_training_callbacks(self, training_callback_attributes):
            """Step counter (SH ramp) before, statistics + refinement after every training iteration --
            the callbacks a densifying Splatfacto registers."""
            def before(step):
                self.gaussians.step = step

            def after(step):
                if self.refiner is not None:
                    opts = training_callback_attributes.optimizers.optimizers
                    self.refiner.after_train_iteration(step + 1, opts)

            return [TrainingCallback([TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], before),
                    TrainingCallback([TrainingCallbackLocation.AFTER_TRAIN_ITERATION], after)]

        def _camera(self, camera: "Cameras") -> Camera:
            c2w = camera.camera_to_worlds[0].cpu().numpy()
            return Camera.from_c2w_opengl(c2w, float(camera.fx[0]), float(camera.fy[0]), float(camera.cx[0]),
                                          float(camera.cy[0]), int(camera.width[0]), int(camera.height[0]))

        def get_outputs(self, camera):
            self.gaussians.training = bool(self.training)     # nn.Module flag: eval renders at full resolution
            return self.gaussians.render(self._camera(camera))

        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            return self.gaussians.loss_dict(outputs, batch)

        def get_metrics_dict(self, outputs, batch):
            return self.gaussians.metrics_dict(outputs, batch)

        def get_image_metrics_and_images(self, outputs, batch):
            return self.gaussians.image_metrics_and_images(outputs, batch)

    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig

    def _opt(lr, final=None):
        return {"optimizer": AdamOptimizerConfig(lr=lr, eps=1e-15),
                "scheduler": ExponentialDecaySchedulerConfig(lr_final=final, max_steps=30000) if final else None}

    depth_gaussian_splatting = MethodSpecification(
        config=TrainerConfig(
            method_name=METHOD_NAME,
            steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000, mixed_precision=False,
            pipeline=VanillaPipelineConfig(
                # the dataset type rides on the datamanager's generic parameter, as in the reference's
                # legacy plugin (config_tactile.py:32-33: VanillaDataManager[DepthDataset])
                datamanager=FullImageDatamanagerConfig(_target=FullImageDatamanager[TactileDepthDataset],
                                                       dataparser=TactileDataParserConfig(load_3D_points=True)),
                model=DepthGSModelConfig()),
            optimizers={k: _opt(lr, XYZ_LR_FINAL if k == "xyz" else None) for k, lr in PARAM_GROUP_LRS.items()},
            viewer=ViewerConfig(num_rays_per_chunk=1 << 15), vis="viewer"),
        description=DESCRIPTION)
