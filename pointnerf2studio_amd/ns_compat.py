"""nerfstudio surface used by the plugin: the real classes when nerfstudio is importable, otherwise a
minimal protocol shim with the same names and arithmetic (nerfstudio is not installable in the build
image; the reference's call sites are studio_model.py:14-27,193-224,387-390 and studio_utils.py:1-12).

Shim semantics restated from the public nerfstudio code (un-vendored, un-pinned dependency
`nerfstudio>=0.3.0`, reference pyproject.toml:14):
  MLP(in_dim, num_layers, layer_width, activation, out_activation): `num_layers` Linear layers, hidden
      activation after all but the last, `out_activation` after the last; parameters under `layers.<i>`.
  FieldHead(in_dim, out_dim, activation): one Linear under `net` followed by the activation.
  RGBRenderer(background_color): sum(w * rgb) + bg * (1 - sum(w)); clamped to [0, 1] outside training.
  Model: nn.Module holding `config`, calling populate_modules() from __init__ (which builds the collider),
      forward = collider, then get_outputs; get_outputs_for_camera_ray_bundle = the [H, W] bundle in row-major chunks of
      config.eval_num_rays_per_chunk rays through forward, outputs concatenated and viewed as [H, W, -1] (under no_grad).
  NearFarCollider(near_plane, far_plane, reset_near_plane=True): nears / fars = ones_like(origins[..., :1]) * plane;
      outside training the near plane is 0 unless reset_near_plane is False.
  RayBundle: a dataclass of tensors sharing their leading dims; len() = number of rays; flatten();
      get_row_major_sliced_ray_bundle(start, end) = the flattened bundle's rays [start, end).
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Type

import torch
from torch import nn

try:  # pragma: no cover - exercised only where nerfstudio is installed
    from nerfstudio.cameras.rays import RayBundle
    from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation
    from nerfstudio.field_components.encodings import Encoding
    from nerfstudio.field_components.field_heads import DensityFieldHead, RGBFieldHead
    from nerfstudio.field_components.mlp import MLP
    from nerfstudio.model_components.losses import MSELoss
    from nerfstudio.model_components.renderers import RGBRenderer
    from nerfstudio.models.base_model import Model, ModelConfig
    HAVE_NERFSTUDIO = True
except Exception:  # ModuleNotFoundError in the build image
    HAVE_NERFSTUDIO = False

    @dataclass
    class RayBundle:
        origins: torch.Tensor
        directions: torch.Tensor
        nears: Optional[torch.Tensor] = None
        fars: Optional[torch.Tensor] = None
        metadata: Dict[str, torch.Tensor] = field(default_factory=dict)
        camera_indices: Optional[torch.Tensor] = None

        @property
        def shape(self):
            return tuple(self.origins.shape[:-1])

        def __len__(self):
            return self.origins.numel() // self.origins.shape[-1]

        def _map(self, fn) -> "RayBundle":
            f = lambda t: None if t is None else fn(t)
            return RayBundle(origins=fn(self.origins), directions=fn(self.directions), nears=f(self.nears),
                             fars=f(self.fars), metadata={k: fn(v) for k, v in self.metadata.items()},
                             camera_indices=f(self.camera_indices))

        def flatten(self) -> "RayBundle":
            lead = len(self.origins.shape) - 1
            return self._map(lambda t: t.reshape((-1,) + tuple(t.shape[lead:])))

        def __getitem__(self, idx) -> "RayBundle":
            return self._map(lambda t: t[idx])

        def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
            return self.flatten()[start_idx:end_idx]

    class TrainingCallbackLocation:
        BEFORE_TRAIN_ITERATION = "before_train_iteration"
        AFTER_TRAIN_ITERATION = "after_train_iteration"
        AFTER_TRAIN = "after_train"

    @dataclass
    class TrainingCallbackAttributes:
        optimizers: Any = None
        grad_scaler: Any = None
        pipeline: Any = None
        trainer: Any = None

    class TrainingCallback:
        def __init__(self, where_to_run, func: Callable, update_every_num_iters: Optional[int] = None,
                     iters=None, args: Optional[List] = None, kwargs: Optional[Dict] = None):
            self.where_to_run = where_to_run
            self.func = func
            self.update_every_num_iters = update_every_num_iters
            self.iters = iters
            self.args = args or []
            self.kwargs = kwargs or {}

        def run_callback(self, step: int) -> None:
            if self.update_every_num_iters is None or step % self.update_every_num_iters == 0:
                self.func(*self.args, **self.kwargs, step=step)

        def run_callback_at_location(self, step: int, location) -> None:
            if location in self.where_to_run:
                self.run_callback(step)

    class Encoding(nn.Module):
        def __init__(self, in_dim: int) -> None:
            super().__init__()
            self.in_dim = in_dim

    class MLP(nn.Module):
        def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: Optional[int] = None,
                     activation: Optional[nn.Module] = None, out_activation: Optional[nn.Module] = None) -> None:
            super().__init__()
            self.in_dim, self.num_layers, self.layer_width = in_dim, num_layers, layer_width
            self.out_dim = out_dim if out_dim is not None else layer_width
            self.activation = activation if activation is not None else nn.ReLU()
            self.out_activation = out_activation
            dims = [in_dim] + [layer_width] * (num_layers - 1) + [self.out_dim]
            self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])

        def get_out_dim(self) -> int:
            return self.out_dim

        def forward(self, x):
            for i, layer in enumerate(self.layers):
                x = layer(x)
                if i < len(self.layers) - 1:
                    x = self.activation(x)
            if self.out_activation is not None:
                x = self.out_activation(x)
            return x

    class _FieldHead(nn.Module):
        def __init__(self, in_dim: int, out_dim: int, activation: Optional[nn.Module]) -> None:
            super().__init__()
            self.net = nn.Linear(in_dim, out_dim)
            self.activation = activation

        def forward(self, x):
            x = self.net(x)
            return self.activation(x) if self.activation is not None else x

    class RGBFieldHead(_FieldHead):
        def __init__(self, in_dim: int, activation: Optional[nn.Module] = None) -> None:
            super().__init__(in_dim, 3, activation if activation is not None else nn.Sigmoid())

    class DensityFieldHead(_FieldHead):
        def __init__(self, in_dim: int, activation: Optional[nn.Module] = None) -> None:
            super().__init__(in_dim, 1, activation if activation is not None else nn.Softplus())

    class RGBRenderer(nn.Module):
        def __init__(self, background_color=None) -> None:
            super().__init__()
            self.background_color = background_color

        def forward(self, rgb, weights):
            comp = torch.sum(weights * rgb, dim=-2)
            acc = torch.sum(weights, dim=-2)
            bg = torch.as_tensor(self.background_color, dtype=comp.dtype, device=comp.device)
            comp = comp + bg * (1.0 - acc)
            if not self.training:
                comp = torch.clamp(comp, min=0.0, max=1.0)
            return comp

    MSELoss = nn.MSELoss

    class NearFarCollider(nn.Module):
        def __init__(self, near_plane: float, far_plane: float, reset_near_plane: bool = True) -> None:
            super().__init__()
            self.near_plane, self.far_plane, self.reset_near_plane = near_plane, far_plane, reset_near_plane

        def forward(self, ray_bundle):
            ones = torch.ones_like(ray_bundle.origins[..., 0:1])
            near_plane = self.near_plane if (self.training or not self.reset_near_plane) else 0
            ray_bundle.nears = ones * near_plane
            ray_bundle.fars = ones * self.far_plane
            return ray_bundle

    @dataclass
    class ModelConfig:
        _target: Type = field(default_factory=lambda: Model)
        enable_collider: bool = True
        collider_params: Optional[Dict[str, float]] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
        loss_coefficients: Dict[str, float] = field(default_factory=dict)
        eval_num_rays_per_chunk: int = 4096

        def setup(self, **kwargs):
            return self._target(self, **kwargs)

    class Model(nn.Module):
        config: ModelConfig

        def __init__(self, config, scene_box=None, num_train_data: int = 0, **kwargs) -> None:
            super().__init__()
            self.config = config
            self.scene_box = scene_box
            self.num_train_data = num_train_data
            self.kwargs = kwargs
            self.populate_modules()
            self.device_indicator_param = nn.Parameter(torch.empty(0))

        @property
        def device(self):
            return self.device_indicator_param.device

        def populate_modules(self):
            self.collider = None
            if getattr(self.config, "enable_collider", False):
                assert self.config.collider_params is not None
                self.collider = NearFarCollider(near_plane=self.config.collider_params["near_plane"],
                                                far_plane=self.config.collider_params["far_plane"])

        def get_training_callbacks(self, training_callback_attributes) -> List[TrainingCallback]:
            return []

        def forward(self, ray_bundle):
            if self.collider is not None:
                ray_bundle = self.collider(ray_bundle)
            return self.get_outputs(ray_bundle)

        @torch.no_grad()
        def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle) -> Dict[str, torch.Tensor]:
            num_rays_per_chunk = self.config.eval_num_rays_per_chunk
            image_height, image_width = camera_ray_bundle.origins.shape[:2]
            num_rays = len(camera_ray_bundle)
            outputs_lists = defaultdict(list)
            for i in range(0, num_rays, num_rays_per_chunk):
                ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + num_rays_per_chunk)
                outputs = self.forward(ray_bundle=ray_bundle)
                for output_name, output in outputs.items():
                    if torch.is_tensor(output):
                        outputs_lists[output_name].append(output)
            return {name: torch.cat(chunks).view(image_height, image_width, -1) for name, chunks in outputs_lists.items()}

WHITE = torch.tensor([1.0, 1.0, 1.0])
