"""nerfstudio registration, identical on the outside to the reference's
pointnerf/nerfstudio/studio_{config,pipeline,datamanager}.py:

  entry point   nerfstudio.method_configs: pointnerf2studio = pointnerf2studio_amd.studio_config:pointnerf_original
  method name   "pointnerf-original"                                          (studio_config.py:14)
  optimisers    "fields" Adam 5e-4, "neural_points" Adam 2e-3 (as optim.PointRowAdam: the same update over the rows that
                ever had a gradient), exp decay 0.1 / 1e6 steps                 (studio_config.py:33-48)
  datamanager   one image per batch, `metadata["camrotc2w"]` = c2w[:3,:3]      (studio_datamanager.py:62-110)

The datamanager's logic (one image per batch, the camera rotation in the bundle's metadata) lives in
`PointNerfDataManagerMixin`, which only duck-types its collaborators (image dataloader iterator, pixel sampler, ray
generator, dataset.cameras): it runs -- and is tested, tests/test_studio_config.py -- without nerfstudio.  The classes
that need nerfstudio's bases (VanillaDataManager, VanillaPipeline, TrainerConfig ...) are thin shells around it,
defined only where nerfstudio imports.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import Type

import torch

from torch.optim import lr_scheduler

from .model import PointNerf, PointNerfConfig
from .ns_compat import HAVE_NERFSTUDIO
from .optim import PointRowAdam

METHOD_NAME = "pointnerf-original"
EXPERIMENT_NAME = "pointnerf2studio"
OPTIMIZER_GROUPS = {"fields": 0.0005, "neural_points": 0.002}
EVAL_NUM_RAYS_PER_CHUNK = 2304


def pointnerf_lr_lambda(lr_decay_exp: float = 0.1, lr_decay_iters: int = 1000000):
    """PointNerfScheduler (studio_utils.py:33-44): lr * lr_decay_exp ** (step / lr_decay_iters)."""
    return lambda step: pow(lr_decay_exp, step / lr_decay_iters)


class PointNerfDataManagerMixin:
    """studio_datamanager.py:62-110.  Expects of `self` what nerfstudio's VanillaDataManager provides: `config`
    (random_image_idx), `train_count` / `eval_count`, `iter_{train,eval}_image_dataloader` (iterators of
    {"image_idx": [n], "image": [n,H,W,3]} batches), `{train,eval}_pixel_sampler.sample(batch)`,
    `{train,eval}_ray_generator(indices)`, `{train,eval}_dataset.cameras[...]`.camera_to_worlds, `eval_dataloader`."""

    def _one_image(self, loader_iter, count):
        """One image of the batch: a random one, or image (count - 1) mod n (studio_datamanager.py:66-73,89-96)."""
        image_batch = next(loader_iter)
        n = image_batch["image_idx"].shape[0]
        image_idx = random.randint(0, n - 1) if self.config.random_image_idx else (count - 1) % n
        sel = torch.nonzero(image_batch["image_idx"] == image_idx).squeeze()
        return {"image_idx": torch.tensor(image_idx).unsqueeze(0), "image": image_batch["image"][sel].unsqueeze(0)}

    def next_train(self, step: int):
        self.train_count += 1
        batch = self.train_pixel_sampler.sample(self._one_image(self.iter_train_image_dataloader, self.train_count))
        ray_bundle = self.train_ray_generator(batch["indices"])
        cams = self.train_dataset.cameras[ray_bundle.camera_indices.cpu()]
        ray_bundle.metadata["camrotc2w"] = cams.camera_to_worlds[0][0][0:3, 0:3]
        return ray_bundle, batch

    def next_eval(self, step: int):
        self.eval_count += 1
        # (the reference indexes the eval image by train_count as well, studio_datamanager.py:92)
        batch = self.eval_pixel_sampler.sample(self._one_image(self.iter_eval_image_dataloader, self.train_count))
        ray_bundle = self.eval_ray_generator(batch["indices"])
        cams = self.eval_dataset.cameras[ray_bundle.camera_indices.cpu()]
        ray_bundle.metadata["camrotc2w"] = cams.camera_to_worlds[0][0][0:3, 0:3]
        return ray_bundle, batch

    def next_eval_image(self, step: int):
        for camera_ray_bundle, batch in self.eval_dataloader:
            image_idx = int(camera_ray_bundle.camera_indices[0, 0, 0])
            h, w = camera_ray_bundle.origins.shape[:2]   # the reference hard-codes 800 x 800 (:108)
            rot = self.eval_dataset.cameras[image_idx].camera_to_worlds[0:3, 0:3]
            camera_ray_bundle.metadata["camrotc2w"] = rot[None, None].expand(h, w, -1, -1).reshape(h, w, -1)
            return image_idx, camera_ray_bundle, batch
        raise ValueError("No more eval images")


if HAVE_NERFSTUDIO:  # pragma: no cover - nerfstudio is not installable in the build image; the block runs against a
    # stand-in package in tests/test_studio_config_registration.py
    import typing

    from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager, VanillaDataManagerConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig
    from nerfstudio.engine.schedulers import Scheduler, SchedulerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.pipelines.base_pipeline import DDP, Model, Pipeline, VanillaPipeline, VanillaPipelineConfig, dist
    from nerfstudio.plugins.types import MethodSpecification

    @dataclass
    class PointNerfSchedulerConfig(SchedulerConfig):
        _target: Type = field(default_factory=lambda: PointNerfScheduler)
        lr_decay_iters: int = 1000000
        lr_decay_exp: float = 0.1

    class PointNerfScheduler(Scheduler):
        config: PointNerfSchedulerConfig

        def get_scheduler(self, optimizer, lr_init: float):
            return lr_scheduler.LambdaLR(
                optimizer, lr_lambda=pointnerf_lr_lambda(self.config.lr_decay_exp, self.config.lr_decay_iters))

    @dataclass
    class PointNerfDataManagerConfig(VanillaDataManagerConfig):
        _target: Type = field(default_factory=lambda: PointNerfDataManager)
        random_image_idx: bool = True
        near_plane: float = 2.0
        far_plane: float = 6.0

    class PointNerfDataManager(PointNerfDataManagerMixin, VanillaDataManager):
        """One image per batch; adds the camera rotation the model needs (studio_datamanager.py:62-110)."""
        config: PointNerfDataManagerConfig

    class PointNerfPipeline(VanillaPipeline):
        """studio_pipeline.py:16-53: hands the cameras to the model, wraps in DDP when world_size > 1."""

        def __init__(self, config, device: str, test_mode="val", world_size: int = 1, local_rank: int = 0,
                     grad_scaler=None):
            Pipeline.__init__(self)
            self.config = config
            self.test_mode = test_mode
            self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size,
                                                        local_rank=local_rank)
            self.datamanager.to(device)
            assert self.datamanager.train_dataset is not None, "Missing input dataset"
            self._model = config.model.setup(scene_box=self.datamanager.train_dataset.scene_box,
                                             num_train_data=len(self.datamanager.train_dataset),
                                             cameras=self.datamanager.train_dataset.cameras)
            self.model.to(device)
            self.world_size = world_size
            if world_size > 1:
                # the reference's wrap (studio_pipeline.py:48-53) with the point tensors taken out of DDP's dense
                # all-reduce: the fused backward exchanges their touched rows itself (distributed.wrap_data_parallel)
                from .distributed import wrap_data_parallel
                self._model = typing.cast(Model, wrap_data_parallel(self._model, device_ids=[local_rank],
                                                                    find_unused_parameters=True))
                dist.barrier(device_ids=[local_rank])

    def _opt(name, lr):
        # the reference: AdamOptimizerConfig(lr) for both groups (studio_config.py:33-48).  The point tensors take the
        # same Adam through optim.PointRowAdam -- torch.optim.Adam's update applied to the rows that ever had a gradient
        # (all others have zero moments and would move by exactly 0): no O(N) sweep per step
        target = {"_target": PointRowAdam} if name == "neural_points" else {}
        return {"optimizer": AdamOptimizerConfig(lr=lr, **target),
                "scheduler": PointNerfSchedulerConfig(lr_decay_exp=0.1, lr_decay_iters=1000000)}

    pointnerf_config = TrainerConfig(
        method_name=METHOD_NAME,
        experiment_name=EXPERIMENT_NAME,
        pipeline=VanillaPipelineConfig(
            _target=PointNerfPipeline,
            datamanager=PointNerfDataManagerConfig(_target=PointNerfDataManager, eval_num_rays_per_batch=4096,
                                                   train_num_rays_per_batch=4096),
            # (PointNerfDataManager hands over one image per batch: every bundle is one camera)
            model=PointNerfConfig(_target=PointNerf, eval_num_rays_per_chunk=EVAL_NUM_RAYS_PER_CHUNK,
                                  hip_single_camera_bundles=True),
        ),
        max_num_iterations=200000,
        steps_per_save=25000,
        steps_per_eval_batch=1000,
        steps_per_eval_image=2000,
        steps_per_eval_all_images=100000,
        optimizers={name: _opt(name, lr) for name, lr in OPTIMIZER_GROUPS.items()},
    )
    pointnerf_original = MethodSpecification(config=pointnerf_config,
                                             description="Point-NeRF for nerfstudio, MI355X-native render path.")
else:
    pointnerf_original = None
