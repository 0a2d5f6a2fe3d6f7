"""nerfstudio method plugin ``depth-gaussian-splatting`` (import-guarded).

The reference trains through a nerfstudio fork's method of this name
(scripts/train_bunny_real.sh:52); the only in-tree evidence of how Touch-GS plugs into nerfstudio
is the legacy NeRF plugin (legacy/config_tactile.py:23-56: MethodSpecification(TrainerConfig(
method_name, pipeline=VanillaPipelineConfig(datamanager, model), optimizers), description)), which
this file mirrors.  nerfstudio is not installable here (no network), so the MethodSpecification and
the ``Model`` shell are only built when it imports and have never executed; everything the shell
delegates to (``AutogradGaussians``: parameter groups, differentiable render, loss with nerfstudio
batch shapes) is free of nerfstudio imports and IS exercised by the GPU tests
(tests/test_gpu_api_surfaces.py::test_nerfstudio_adapter_core_trains).  ``touch_gs_amd.train`` is
the self-contained trainer (fused path).

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

    def param_groups(self):
        return {k: [v] for k, v in self.params.items()}

    def render(self, cam, sh_degree=None):
        import torch
        from . import ops
        p = self.params
        sh = torch.cat([p["features_dc"], p["features_rest"]], dim=1)
        rgb, depth_acc, alpha, radii = ops.render(p["xyz"], p["scaling"], p["rotation"], p["opacity"], sh, cam,
                                                  self.sh_degree if sh_degree is None else sh_degree,
                                                  budget=self.core._sync_budget)
        depth = depth_acc / torch.clamp(alpha, min=1e-10)
        return dict(rgb=rgb, depth=depth[..., None], accumulation=alpha[..., None], depth_acc=depth_acc, alpha=alpha,
                    radii=radii)

    @staticmethod
    def view_from_batch(batch, like):
        """nerfstudio batch -> View: depth_image / uncertainty arrive as [H,W,1]."""
        from .model import View
        sq = lambda t: None if t is None else t.to(like.device).reshape(like.shape[0], like.shape[1]).float()
        return View(cam=None, rgb=batch["image"].to(like).float(), depth=sq(batch.get("depth_image")),
                    uncertainty=sq(batch.get("uncertainty")))

    def loss_dict(self, outputs, batch):
        return self.core.get_loss_dict(outputs, self.view_from_batch(batch, outputs["rgb"]))

    def metrics_dict(self, outputs, batch):
        return self.core.get_metrics_dict(outputs, self.view_from_batch(batch, outputs["rgb"]))

    def image_metrics_and_images(self, outputs, batch):
        return self.core.get_image_metrics_and_images(outputs, self.view_from_batch(batch, outputs["rgb"]))


if available:  # pragma: no cover
    from dataclasses import dataclass, field
    from typing import Type

    from nerfstudio.cameras.cameras import Cameras
    from nerfstudio.data.datamanagers.full_images_datamanager import FullImageDatamanagerConfig
    from nerfstudio.data.dataparsers.nerfstudio_dataparser import NerfstudioDataParserConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig
    from nerfstudio.models.base_model import Model, ModelConfig as NSModelConfig
    from nerfstudio.pipelines.base_pipeline import VanillaPipelineConfig

    from .camera import Camera
    from .model import DepthGaussianSplattingModel, ModelConfig, View
    from .optim import GaussianParams

    @dataclass
    class DepthGSModelConfig(NSModelConfig):
        _target: Type = field(default_factory=lambda: DepthGSNerfstudioModel)
        depth_loss_mult: float = 0.2
        depth_loss_type: str = "DEPTH_UNCERTAINTY_WEIGHTED_LOSS"
        uncertainty_weight: float = 1.0
        sh_degree: int = 3
        ssim_lambda: float = 0.2
        num_random: int = 50000

    class DepthGSNerfstudioModel(Model):
        """Thin shell: nerfstudio Model API -> AutogradGaussians (above).  Parameters are nn.Parameters
        in Splatfacto's six groups, so nerfstudio's loss.backward() + optimizer steps train them."""
        config: DepthGSModelConfig

        def populate_modules(self):
            import torch
            seed = self.kwargs.get("seed_points")
            if seed is not None:
                means, cols = seed[0].float(), seed[1].float() / 255.0
            else:
                means = (torch.rand(self.config.num_random, 3) - 0.5) * 2
                cols = torch.rand(self.config.num_random, 3)
            self.gaussians = AutogradGaussians(self.config, means, cols)
            self.gauss_params = torch.nn.ParameterDict(self.gaussians.params)   # registered: state_dict, .to()

        def get_param_groups(self):
            return self.gaussians.param_groups()

        def _camera(self, camera: "Cameras") -> Camera:
            c2w = camera.camera_to_worlds[0].cpu().numpy()
            return Camera.from_c2w_opengl(c2w, float(camera.fx[0]), float(camera.fy[0]), float(camera.cx[0]),
                                          float(camera.cy[0]), int(camera.width[0]), int(camera.height[0]))

        def get_outputs(self, camera):
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
                datamanager=FullImageDatamanagerConfig(dataparser=NerfstudioDataParserConfig(load_3D_points=True)),
                model=DepthGSModelConfig()),
            optimizers={k: _opt(lr, XYZ_LR_FINAL if k == "xyz" else None) for k, lr in PARAM_GROUP_LRS.items()},
            viewer=ViewerConfig(num_rays_per_chunk=1 << 15), vis="viewer"),
        description=DESCRIPTION)
