"""Executed as a script by tests/test_studio_config_registration.py (its own process: it rewires sys.modules).

nerfstudio is not installable in the build image, so the part of pointnerf2studio_amd/studio_config.py that needs its base
classes (the TrainerConfig / MethodSpecification registration, the datamanager and pipeline shells) never runs in the other
tests.  Here a stand-in `nerfstudio` package is assembled in sys.modules -- the model-side names are the protocol shim's
own classes (ns_compat), the trainer-side names are plain dataclasses with the public nerfstudio field names the
registration uses -- the package is imported afresh, and the registration is checked against what the reference registers
(studio_config.py:14-50, studio_pipeline.py:16-53, studio_datamanager.py:41-60).  A stand-in proves the block imports,
constructs and wires what it says; it is not a substitute for running under the real nerfstudio.
"""
import dataclasses
import importlib
import os
import sys
import types
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Type

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.environ.get("PNR_TEST_INSTALLED_COPY"):     # (tests/test_packaging.py imports an INSTALLED copy instead)
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch import nn  # noqa: E402


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        if parent not in sys.modules:
            _module(parent)
        setattr(sys.modules[parent], leaf, m)
    return m


def install_stand_in():
    shim = importlib.import_module("pointnerf2studio_amd.ns_compat")
    assert not shim.HAVE_NERFSTUDIO
    _module("nerfstudio")
    _module("nerfstudio.cameras.rays", RayBundle=shim.RayBundle)
    _module("nerfstudio.engine.callbacks", TrainingCallback=shim.TrainingCallback,
            TrainingCallbackAttributes=shim.TrainingCallbackAttributes,
            TrainingCallbackLocation=shim.TrainingCallbackLocation)
    _module("nerfstudio.field_components.encodings", Encoding=shim.Encoding)
    _module("nerfstudio.field_components.field_heads", DensityFieldHead=shim.DensityFieldHead,
            RGBFieldHead=shim.RGBFieldHead)
    _module("nerfstudio.field_components.mlp", MLP=shim.MLP)
    _module("nerfstudio.model_components.losses", MSELoss=shim.MSELoss)
    _module("nerfstudio.model_components.renderers", RGBRenderer=shim.RGBRenderer)
    _module("nerfstudio.model_components.scene_colliders", NearFarCollider=shim.NearFarCollider)
    _module("nerfstudio.models.base_model", Model=shim.Model, ModelConfig=shim.ModelConfig)

    # ---- trainer side: plain dataclasses with the field names of nerfstudio's public configs --------------------------
    @dataclass
    class InstantiateConfig:
        _target: Type = None

        def setup(self, **kwargs):
            return self._target(self, **kwargs)

    @dataclass
    class VanillaDataManagerConfig(InstantiateConfig):
        train_num_rays_per_batch: int = 1024
        eval_num_rays_per_batch: int = 1024

    class VanillaDataManager(nn.Module):
        def __init__(self, config, device="cpu", test_mode="val", world_size=1, local_rank=0, **kwargs):
            super().__init__()
            self.config = config
            self.train_count = self.eval_count = 0
            cams = types.SimpleNamespace(camera_to_worlds=torch.eye(4)[None, :3])
            self.train_dataset = _Dataset(cams)

    class _Dataset:
        def __init__(self, cams):
            self.scene_box, self.cameras = "box", cams

        def __len__(self):
            return 7

    @dataclass
    class AdamOptimizerConfig:      # nerfstudio.engine.optimizers [ns-mem]: _target, lr, eps, max_norm, weight_decay
        _target: Type = torch.optim.Adam
        lr: float = 1e-3
        eps: float = 1e-8
        max_norm: Optional[float] = None
        weight_decay: float = 0

        def setup(self, params):
            kwargs = vars(self).copy()
            kwargs.pop("_target")
            kwargs.pop("max_norm")
            return self._target(params, **kwargs)

    @dataclass
    class SchedulerConfig(InstantiateConfig):
        pass

    class Scheduler:
        def __init__(self, config):
            self.config = config

    @dataclass
    class VanillaPipelineConfig(InstantiateConfig):
        datamanager: Any = None
        model: Any = None

    class Pipeline(nn.Module):
        @property
        def model(self):
            return self._model

    class VanillaPipeline(Pipeline):
        pass

    @dataclass
    class TrainerConfig:
        method_name: str = ""
        experiment_name: str = ""
        pipeline: Any = None
        max_num_iterations: int = 0
        steps_per_save: int = 0
        steps_per_eval_batch: int = 0
        steps_per_eval_image: int = 0
        steps_per_eval_all_images: int = 0
        optimizers: Dict[str, Any] = field(default_factory=dict)

    @dataclass
    class MethodSpecification:
        config: Any = None
        description: str = ""

    _module("nerfstudio.data.datamanagers.base_datamanager", VanillaDataManager=VanillaDataManager,
            VanillaDataManagerConfig=VanillaDataManagerConfig)
    _module("nerfstudio.engine.optimizers", AdamOptimizerConfig=AdamOptimizerConfig)
    _module("nerfstudio.engine.schedulers", Scheduler=Scheduler, SchedulerConfig=SchedulerConfig)
    _module("nerfstudio.engine.trainer", TrainerConfig=TrainerConfig)
    _module("nerfstudio.pipelines.base_pipeline", DDP=torch.nn.parallel.DistributedDataParallel, Model=shim.Model,
            Pipeline=Pipeline, VanillaPipeline=VanillaPipeline, VanillaPipelineConfig=VanillaPipelineConfig,
            dist=torch.distributed)
    _module("nerfstudio.plugins.types", MethodSpecification=MethodSpecification)
    # the package again, this time finding `nerfstudio`
    for name in [n for n in sys.modules if n == "pointnerf2studio_amd" or n.startswith("pointnerf2studio_amd.")]:
        del sys.modules[name]


def main():
    install_stand_in()
    ns = importlib.import_module("pointnerf2studio_amd.ns_compat")
    assert ns.HAVE_NERFSTUDIO, "the stand-in package was not picked up"
    sc = importlib.import_module("pointnerf2studio_amd.studio_config")
    spec = sc.pointnerf_original
    assert spec is not None and spec.config.method_name == "pointnerf-original"           # studio_config.py:14
    cfg = spec.config
    assert cfg.experiment_name == "pointnerf2studio"
    assert set(cfg.optimizers) == {"fields", "neural_points"}                                # studio_config.py:33-48
    assert cfg.optimizers["fields"]["optimizer"].lr == 0.0005 and cfg.optimizers["neural_points"]["optimizer"].lr == 0.002
    # both groups are Adam; the point tensors' through the row-sparse form of the same update, constructed the way
    # nerfstudio's Optimizers does (config.setup(params))
    from pointnerf2studio_amd.optim import PointRowAdam
    assert cfg.optimizers["fields"]["optimizer"]._target is torch.optim.Adam
    assert cfg.optimizers["neural_points"]["optimizer"]._target is PointRowAdam
    o = cfg.optimizers["neural_points"]["optimizer"].setup([nn.Parameter(torch.zeros(1, 5, 3))])
    assert isinstance(o, PointRowAdam) and o.defaults["lr"] == 0.002 and o.defaults["eps"] == 1e-8
    assert o.defaults["betas"] == (0.9, 0.999)
    for group in cfg.optimizers.values():
        sch = group["scheduler"]
        assert (sch.lr_decay_exp, sch.lr_decay_iters) == (0.1, 1000000)
        opt = torch.optim.Adam([nn.Parameter(torch.zeros(1))], lr=1.0)
        lam = sch.setup().get_scheduler(opt, 1.0)                                            # studio_utils.py:33-44
        for _ in range(3):
            opt.step()
            lam.step()
        assert abs(lam.get_last_lr()[0] - 0.1 ** (3 / 1000000)) < 1e-12
    assert (cfg.max_num_iterations, cfg.steps_per_save, cfg.steps_per_eval_batch, cfg.steps_per_eval_image,
            cfg.steps_per_eval_all_images) == (200000, 25000, 1000, 2000, 100000)          # studio_config.py:17-22
    dm = cfg.pipeline.datamanager
    assert dm._target is sc.PointNerfDataManager and dm.random_image_idx is True            # studio_datamanager.py:41-60
    assert (dm.train_num_rays_per_batch, dm.eval_num_rays_per_batch, dm.near_plane, dm.far_plane) == (4096, 4096, 2.0, 6.0)
    assert issubclass(sc.PointNerfDataManager, sc.PointNerfDataManagerMixin)
    mc = cfg.pipeline.model
    assert mc._target.__name__ == "PointNerf" and mc.eval_num_rays_per_chunk == 2304          # studio_config.py:25
    assert mc.hip_single_camera_bundles is True
    assert cfg.pipeline._target is sc.PointNerfPipeline

    # the pipeline shell (studio_pipeline.py:16-53) with a stand-in model config: the datamanager is set up and moved, the
    # model receives the scene box, the image count and the cameras, world_size 1 wraps nothing
    seen = {}

    class TinyModel(nn.Module):
        def __init__(self, config, **kwargs):
            super().__init__()
            seen.update(kwargs)
            self.w = nn.Parameter(torch.zeros(1))

    pcfg = dataclasses.replace(cfg.pipeline, model=types.SimpleNamespace(setup=lambda **kw: TinyModel(None, **kw)))
    pipe = pcfg.setup(device="cpu", test_mode="val", world_size=1, local_rank=0)
    assert isinstance(pipe.datamanager, sc.PointNerfDataManager) and pipe.world_size == 1
    assert seen["scene_box"] == "box" and seen["num_train_data"] == 7 and seen["cameras"] is pipe.datamanager.train_dataset.cameras
    assert isinstance(pipe.model, TinyModel)
    print("registration ok")


if __name__ == "__main__":
    main()
