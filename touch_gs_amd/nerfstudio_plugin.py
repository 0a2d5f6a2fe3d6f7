"""nerfstudio method plugin ``depth-gaussian-splatting`` (import-guarded).

The reference trains through a nerfstudio fork's method of this name
(scripts/train_bunny_real.sh:52); the only in-tree evidence of how Touch-GS plugs into nerfstudio
is the legacy NeRF plugin (legacy/config_tactile.py:23-56: MethodSpecification(TrainerConfig(
method_name, pipeline=VanillaPipelineConfig(datamanager, model), optimizers), description)), which
this file mirrors.  nerfstudio is not installable here (no network), so the spec is only built when
it imports; ``touch_gs_amd.train`` is the self-contained trainer that is actually exercised.

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
        """Adapter: nerfstudio Model API -> touch_gs_amd.model (autograd path)."""
        config: DepthGSModelConfig

        def populate_modules(self):
            import torch
            seed = self.kwargs.get("seed_points")
            dev = "cuda"
            if seed is not None:
                means = seed[0].float().to(dev)
                cols = seed[1].float().to(dev) / 255.0
            else:
                means = (torch.rand(self.config.num_random, 3, device=dev) - 0.5) * 2
                cols = torch.rand(self.config.num_random, 3, device=dev)
            N, K = means.shape[0], (self.config.sh_degree + 1) ** 2
            sh = torch.zeros(N, K, 3, device=dev)
            sh[:, 0] = (cols - 0.5) / 0.28209479177387814
            gp = GaussianParams.from_tensors(means, torch.full((N, 3), -4.0, device=dev),
                                             torch.nn.functional.normalize(torch.randn(N, 4, device=dev)),
                                             torch.full((N,), -2.0, device=dev), sh)
            cfg = ModelConfig(sh_degree=self.config.sh_degree, ssim_lambda=self.config.ssim_lambda,
                              depth_loss_mult=self.config.depth_loss_mult, depth_loss_type=self.config.depth_loss_type,
                              uncertainty_weight=self.config.uncertainty_weight)
            self.core = DepthGaussianSplattingModel(cfg, gp)

        def get_param_groups(self):
            return {}

        def _camera(self, camera: "Cameras") -> Camera:
            c2w = camera.camera_to_worlds[0].cpu().numpy()
            return Camera.from_c2w_opengl(c2w, float(camera.fx[0]), float(camera.fy[0]), float(camera.cx[0]),
                                          float(camera.cy[0]), int(camera.width[0]), int(camera.height[0]))

        def get_outputs(self, camera):
            return self.core.get_outputs(self._camera(camera))

        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            view = View(cam=None, rgb=batch["image"].to(outputs["rgb"]), depth=batch.get("depth_image"),
                        uncertainty=batch.get("uncertainty"))
            return self.core.get_loss_dict(outputs, view)

        def get_metrics_dict(self, outputs, batch):
            view = View(cam=None, rgb=batch["image"].to(outputs["rgb"]), depth=batch.get("depth_image"))
            return self.core.get_metrics_dict(outputs, view)

        def get_image_metrics_and_images(self, outputs, batch):
            view = View(cam=None, rgb=batch["image"].to(outputs["rgb"]), depth=batch.get("depth_image"))
            return self.core.get_image_metrics_and_images(outputs, view)

    depth_gaussian_splatting = MethodSpecification(
        config=TrainerConfig(
            method_name=METHOD_NAME,
            steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000, mixed_precision=False,
            pipeline=VanillaPipelineConfig(
                datamanager=FullImageDatamanagerConfig(dataparser=NerfstudioDataParserConfig(load_3D_points=True)),
                model=DepthGSModelConfig()),
            optimizers={"dummy": {"optimizer": AdamOptimizerConfig(lr=1e-3), "scheduler": None}},
            viewer=ViewerConfig(num_rays_per_chunk=1 << 15), vis="viewer"),
        description=DESCRIPTION)
