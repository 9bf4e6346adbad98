"""Host-side mirror of the reference's nerfstudio Model (pointnerf/nerfstudio/studio_model.py):
`PointNerfConfig` with the same fields and defaults, `PointNerf` with the same module names (so state
dicts and optimiser groups carry over), `get_outputs` / `get_param_groups` / `get_loss_dict` /
`get_training_callbacks` / `fill_invalid` / `linear`.

get_outputs has two bodies:
  * outside training (the path the metric times, `get_outputs_for_camera_ray_bundle`): ONE call into the
    fused HIP renderer (pnr_render) -- query, gather, MLPs on fp32 MFMA, composite;
  * in training mode: the reference's own op sequence on PyTorch-ROCm tensors (autograd needs it until the
    backward kernels of SURVEY.md 8f-1 exist), with the HIP drop-in op doing the query.
There is no CPU path in either.
"""
from __future__ import annotations

import dataclasses
import glob
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn
from torch.nn import Parameter

from . import metrics
from .neural_points import NeuralPoints, PointNeRFEncoding
from .ns_compat import (MLP, DensityFieldHead, Model, ModelConfig, MSELoss, RGBFieldHead, RGBRenderer,
                        TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation, WHITE)
from ._lib import MAX_CAMS
from .renderer import MLP_TENSOR_ORDER, RendererHIP, WeightsHIP


def get_latest_epoch(resume_dir):
    """studio_model.py:55-59."""
    os.makedirs(resume_dir, exist_ok=True)
    str_epoch = [f.split("_")[0] for f in os.listdir(resume_dir) if f.endswith("_states.pth")]
    int_epoch = [int(i) for i in str_epoch]
    return None if len(int_epoch) == 0 else str_epoch[int_epoch.index(max(int_epoch))]


@dataclass
class PointNerfConfig(ModelConfig):
    """Field-for-field the reference's PointNerfConfig (studio_model.py:61-118)."""
    _target: Any = dataclasses.field(default_factory=lambda: PointNerf)
    path_point_cloud: Optional[Path] = None
    eval_num_rays_per_chunk: int = 4096

    feat_grad: bool = True
    conf_grad: bool = True
    dir_grad: bool = True
    color_grad: bool = True

    num_pos_freqs: Optional[int] = 10
    num_viewdir_freqs: Optional[int] = 4
    num_feat_freqs: Optional[int] = 3
    num_dist_freqs: Optional[int] = 5

    agg_dist_pers: Optional[int] = 20
    point_features_dim: Optional[int] = 32

    point_color_mode: Optional[bool] = True
    point_dir_mode: Optional[bool] = True

    num_samples: int = 80
    use_biased_sampler: bool = False
    field_dim: int = 64

    num_mlp_base_layers: Optional[int] = 2
    num_mlp_head_layers: Optional[int] = 2
    num_color_layers: Optional[int] = 3
    num_alpha_layers: Optional[int] = 1
    hidden_size: int = 256
    hidden_size_color: int = 128

    apply_pnt_mask: bool = True
    act_super: bool = False
    axis_weight: List[float] = dataclasses.field(default_factory=lambda: [1., 1., 1.])
    kernel_size: List[int] = dataclasses.field(default_factory=lambda: [3, 3, 3])
    vscale: List[float] = dataclasses.field(default_factory=lambda: [2, 2, 2])
    vsize: List[float] = dataclasses.field(default_factory=lambda: [0.004, 0.004, 0.004])
    query_size: List[float] = dataclasses.field(default_factory=lambda: [3, 3, 3])
    ranges: List[float] = dataclasses.field(default_factory=lambda: [-1.200, -1.200, -1.200, 1.200, 1.200, 1.200])
    z_depth_dim: int = 400

    SR: int = 80
    K: int = 8
    max_o: int = 1000000
    P: int = 12
    NN: int = 2
    gpu_maxthr: int = 1024

    zero_epsilon: float = 1e-3
    zero_one_loss_weights: float = 0.0001

    # additions of this build (not in the reference): arithmetic of the fused HIP MLP, see include/pnr.h
    # "fp32" (default: every product and sum in fp32, the reference's arithmetic) or the opt-in fast mode "bf16x3"
    # (3 bf16 MFMA products per fp32 product: image within ~2e-5 of fp32, gradients with ~1e-2 relative noise)
    hip_mlp_mode: str = "fp32"
    hip_early_stop_eps: float = 0.0  # eval only: > 0 stops shading a ray once its transmittance is below eps
    hip_fused_training: bool = True  # training: fused HIP render + pnr_render_backward instead of torch autograd
    # opt-in (a behaviour change: the reference discards them, studio_utils.py:84-90): initialise the plugin MLPs from
    # the `aggregator.*` tensors of the legacy checkpoint (same layer shapes; the legacy net was trained with
    # LeakyReLU slope 0.01 and a Softplus density, so this is a warm start, not an equivalence)
    hip_load_aggregator_weights: bool = False

    def __post_init__(self):
        if self.path_point_cloud is not None:
            if not Path(self.path_point_cloud).exists():
                raise RuntimeError(f"PointCloud path {self.path_point_cloud} does not exist")


class _FusedRenderFn(torch.autograd.Function):
    """pnr_render_views forwards, pnr_render_backward backwards (include/pnr.h).  Inputs after `ray_cam`: points_embeding,
    points_color, points_dir and the nine (weight, bias) pairs in MLP_TENSOR_ORDER."""

    @staticmethod
    def forward(ctx, rnd, dirs, cams, ray_cam, emb, color, pdir, *mlp):
        out = rnd.render_views(dirs, cams, dirs.reshape(-1, 3).shape[0], ray_cam=ray_cam)
        rnd.last_counters = out["counters"]
        ctx.rnd = rnd
        ctx.shapes = (emb.shape, color.shape, pdir.shape)
        ctx.state = {name + suf: mlp[2 * i + j] for i, name in enumerate(MLP_TENSOR_ORDER)
                     for j, suf in enumerate((".weight", ".bias"))}
        ctx.call = rnd.calls
        ctx.mark_non_differentiable(out["ray_mask"])
        return out["rgb"], out["ray_mask"]

    @staticmethod
    def backward(ctx, g_rgb, _g_mask):
        rnd = ctx.rnd
        if rnd.calls != ctx.call:
            raise RuntimeError("fused training: the renderer ran another render before backward(); its workspace "
                               "no longer holds this step's sample lists")
        es, cs, ds = ctx.shapes
        g = rnd.backward(g_rgb, ctx.state, es[-2])
        grads = [g["embedding"].view(es), g["color"].view(cs), g["dir"].view(ds)]
        for name in MLP_TENSOR_ORDER:
            grads += [g[name + ".weight"], g[name + ".bias"]]
        return (None, None, None, None, *grads)


class PointNerf(Model):
    """studio_model.py:121-505."""
    config: PointNerfConfig

    def __init__(self, config: PointNerfConfig, cameras=None, point_state_dict: Optional[Dict] = None,
                 **kwargs) -> None:
        self._point_state_dict = point_state_dict   # lets tests / benches hand the cloud over in memory
        super().__init__(config=config, **kwargs)
        self._point_initialized = False
        self.cameras = cameras
        self._device = "cuda"
        self._renderer: Optional[RendererHIP] = None
        self._renderer_train: Optional[RendererHIP] = None
        self._weights: Optional[WeightsHIP] = None
        self._weights_key = None
        self._render_calls = 0
        self._init_pointnerf()

    def _init_pointnerf(self):
        """studio_model.py:147-166: newest `{iter}_net_ray_marching.pth` of path_point_cloud; only the
        `neural_points.*` keys are consumed (studio_utils.py:84-90)."""
        if self._point_state_dict is not None:
            state_dict = self._point_state_dict
        elif self.config.path_point_cloud is not None:
            path = str(self.config.path_point_cloud)
            if not os.path.exists(path):
                raise RuntimeError(f"Specified point_cloud path {path} does not exist")
            if len([n for n in glob.glob(path + "/*_net_ray_marching.pth") if os.path.isfile(n)]) == 0:
                raise RuntimeError(f"Cannot find any _net_ray_marching.pth in {path}")
            load_path = os.path.join(path, '{}_net_ray_marching.pth'.format(get_latest_epoch(path)))
            if not os.path.isfile(load_path):
                raise RuntimeError(f'cannot load {load_path}')
            state_dict = torch.load(load_path, map_location="cpu")
        else:
            raise RuntimeError("The point_cloud_path must be specified.")
        self.neural_points = NeuralPoints(state_dict, self._device, self.config)
        if getattr(self.config, "hip_load_aggregator_weights", False):
            self.load_aggregator_weights(state_dict)
        self._point_initialized = True

    # legacy PointAggregator module -> plugin module (models/aggregators/point_aggregators.py:193-290 against
    # studio_model.py:193-221; layer shapes are identical, verified in SURVEY.md section 8c)
    AGGREGATOR_MAP = {
        "aggregator.block1.0": "mlp_base.layers.0", "aggregator.block1.2": "mlp_base.layers.1",
        "aggregator.block3.0": "mlp_head.layers.0", "aggregator.block3.2": "mlp_head.layers.1",
        "aggregator.alpha_branch.0": "field_output_density.net",
        "aggregator.color_branch.0": "mlp_color.layers.0", "aggregator.color_branch.2": "mlp_color.layers.1",
        "aggregator.color_branch.4": "mlp_color.layers.2", "aggregator.color_branch.6": "field_output_color.net",
    }

    def load_aggregator_weights(self, state_dict: Dict[str, torch.Tensor]) -> int:
        """Copies the nine Linear layers of a legacy checkpoint's `aggregator.*` into the plugin MLPs; every tensor must
        be present with the plugin's shape (nothing is partially loaded).  Returns the number of tensors copied."""
        todo = []
        for src, dst in self.AGGREGATOR_MAP.items():
            mod = self.get_submodule(dst)
            for suf, param in ((".weight", mod.weight), (".bias", mod.bias)):
                if src + suf not in state_dict:
                    raise RuntimeError(f"hip_load_aggregator_weights: checkpoint has no {src + suf}")
                t = state_dict[src + suf]
                if tuple(t.shape) != tuple(param.shape):
                    raise RuntimeError(f"hip_load_aggregator_weights: {src + suf} has shape {tuple(t.shape)}, "
                                       f"{dst + suf} needs {tuple(param.shape)}")
                todo.append((param, t))
        with torch.no_grad():
            for param, t in todo:
                param.copy_(t.to(param.dtype))
        self._weights_key = None
        return len(todo)

    def populate_modules(self):
        """studio_model.py:169-237."""
        super().populate_modules()
        cfg = self.config
        self.direction_encoding = PointNeRFEncoding(in_dim=2, num_frequencies=cfg.num_viewdir_freqs, ori=True)
        self.feature_encoding = PointNeRFEncoding(in_dim=2, num_frequencies=cfg.num_feat_freqs, ori=False)
        self.dists_encoding = PointNeRFEncoding(in_dim=2, num_frequencies=cfg.num_dist_freqs, ori=False)
        dist_dim = (4 if cfg.agg_dist_pers == 30 else 6) if cfg.agg_dist_pers > 9 else 3
        dist_xyz_dim = dist_dim if cfg.num_dist_freqs == 0 else 2 * abs(cfg.num_dist_freqs) * dist_dim
        mlp_in_dim = 2 * cfg.num_feat_freqs * cfg.point_features_dim + dist_xyz_dim + cfg.point_features_dim
        act = lambda: nn.LeakyReLU(0.1, True)
        self.mlp_base = MLP(in_dim=mlp_in_dim, num_layers=cfg.num_mlp_base_layers, layer_width=cfg.hidden_size,
                            activation=act(), out_activation=act())
        mlp_in_dim = self.mlp_base.get_out_dim() + (3 if cfg.point_color_mode else 0) + (4 if cfg.point_dir_mode else 0)
        self.mlp_head = MLP(in_dim=mlp_in_dim, num_layers=cfg.num_mlp_head_layers, layer_width=cfg.hidden_size,
                            activation=act(), out_activation=act())
        color_in_dim = self.mlp_head.get_out_dim() + 2 * cfg.num_viewdir_freqs * 3
        self.mlp_color = MLP(in_dim=color_in_dim, num_layers=cfg.num_color_layers, layer_width=cfg.hidden_size_color,
                             activation=act(), out_activation=act())
        self.field_output_color = RGBFieldHead(in_dim=self.mlp_color.get_out_dim(), activation=torch.nn.Sigmoid())
        self.field_output_density = DensityFieldHead(in_dim=self.mlp_head.get_out_dim(), activation=torch.nn.ReLU())
        self._background_color = WHITE
        self.rgb_renderer = RGBRenderer(background_color=self._background_color)
        self.mask_loss = MSELoss()
        self.rgb_loss = MSELoss()
        # metrics (studio_model.py:230-237), same attribute names; see metrics.py for what stands behind each
        self.psnr = metrics.psnr
        self.torchmetrics_ssim = metrics.ssim_gaussian
        self.skimage_ssim = metrics.ssim_uniform
        self.skimage_rmse = metrics.rmse
        self.lpips = metrics.Lpips("alex")
        self.lpips_vgg = metrics.Lpips("vgg")

    # Just to allow for size reduction of the checkpoint (studio_model.py:240-255): the LPIPS networks are never
    # saved, and a checkpoint without them loads under strict=True
    def load_state_dict(self, state_dict, strict: bool = True):
        state_dict = dict(state_dict)
        for name in ("lpips", "lpips_vgg"):
            if hasattr(self, name):
                for k, v in getattr(self, name).state_dict().items():
                    state_dict[f"{name}.{k}"] = v
        return super().load_state_dict(state_dict, strict)

    def state_dict(self, *args, prefix="", **kwargs):
        state_dict = super().state_dict(*args, prefix=prefix, **kwargs)
        for k in list(state_dict.keys()):
            if k.startswith(f"{prefix}lpips.") or k.startswith(f"{prefix}lpips_vgg."):
                state_dict.pop(k)
        return state_dict

    # ---- fused HIP path ---------------------------------------------------------------------------------
    def _fusable(self) -> bool:
        c = self.config
        return (c.point_features_dim == 32 and c.num_feat_freqs == 3 and c.num_dist_freqs == 5 and
                c.num_viewdir_freqs == 4 and c.agg_dist_pers == 20 and c.hidden_size == 256 and
                c.hidden_size_color == 128 and c.num_mlp_base_layers == 2 and c.num_mlp_head_layers == 2 and
                c.num_color_layers == 3 and bool(c.point_color_mode) and bool(c.point_dir_mode) and
                list(c.axis_weight) == [1., 1., 1.] and bool(c.apply_pnt_mask))

    def _mlp_state(self) -> Dict[str, torch.Tensor]:
        sd = {}
        for name in MLP_TENSOR_ORDER:
            mod = self.get_submodule(name)
            sd[name + ".weight"], sd[name + ".bias"] = mod.weight, mod.bias
        return sd

    def _fused_renderer(self, train: bool = False) -> RendererHIP:
        scene = self.neural_points.fused_scene()
        sd = self._mlp_state()
        key = tuple((t.data_ptr(), t._version) for t in sd.values()) + (self.neural_points.points_Rw2c._version,)
        if self._weights is None or key != self._weights_key:
            if self._weights is None:
                self._weights = WeightsHIP()
            self._weights.pack(sd, self.neural_points.points_Rw2c.detach(), self.neural_points.points_xyz.device)
            self._weights_key = key
        c = self.config
        if train:
            # the training renderer has its own workspace and never clamps (nerfstudio's RGBRenderer in training)
            if self._renderer_train is None or self._renderer_train.scene is not scene:
                self._renderer_train = RendererHIP(scene, self._weights, SR=c.SR, K=c.K, D=c.z_depth_dim,
                                                   radius_limit=float(self.neural_points.radius_limit_np),
                                                   vsize_z=c.vsize[2], eval_clamp=False,
                                                   bg=self._background_color.tolist(),
                                                   precision=getattr(c, "hip_mlp_mode", "fp32"))
            self._renderer_train.mlp_state = sd
            return self._renderer_train
        if self._renderer is None or self._renderer.scene is not scene:
            self._renderer = RendererHIP(scene, self._weights, SR=c.SR, K=c.K, D=c.z_depth_dim,
                                         radius_limit=float(self.neural_points.radius_limit_np),
                                         vsize_z=c.vsize[2], eval_clamp=True, bg=self._background_color.tolist(),
                                         precision=getattr(c, "hip_mlp_mode", "fp32"),
                                         early_stop_eps=float(getattr(c, "hip_early_stop_eps", 0.0)))
        return self._renderer

    def _bundle_cameras(self, ray_bundle):
        """The cameras of a bundle.  The reference assumes ONE per bundle and reads origins[0] / camrotc2w[0]
        (studio_utils.py:148-155); so does this for such bundles (one cheap all-equal test).  Bundles that mix cameras
        -- nerfstudio's usual random-pixel batches over several images -- are rendered in
        ONE pnr_render_views call with a per-ray camera index, up to PNR_MAX_CAMS cameras.
        Returns (cams [(pos, rot3x3, near, far)], ray_cam int32 [R] or None)."""
        rot, pos = self.neural_points._camera(ray_bundle)
        near, far = ray_bundle.nears[0].item(), ray_bundle.fars[0].item()
        meta = ray_bundle.metadata["camrotc2w"]
        o = ray_bundle.origins.reshape(-1, 3)
        if meta.shape[0] == 3 or o.shape[0] <= 1:
            return [(pos[0], rot[0], near, far)], None
        key = torch.cat([o, meta.reshape(o.shape[0], 9)], dim=1).to(self._device)
        if bool((key == key[0]).all()):
            return [(pos[0], rot[0], near, far)], None
        uniq, inv = torch.unique(key, dim=0, return_inverse=True)
        if uniq.shape[0] > MAX_CAMS:
            raise RuntimeError(f"a ray bundle may mix at most {MAX_CAMS} cameras, got {uniq.shape[0]}")
        uniq = uniq.cpu()
        cams = [(uniq[i, :3], uniq[i, 3:].view(3, 3), near, far) for i in range(uniq.shape[0])]
        return cams, inv.to(torch.int32)

    def _get_outputs_fused(self, ray_bundle):
        """Jitter: the reference draws torch.rand jitter even at eval (studio_utils.py:166, hard-coded 0.3); the
        fused path uses the same fraction (`neural_points.jitter`) with the library's counter-based uniforms and
        a fresh seed per call.  Set `neural_points.jitter = 0` for deterministic mid-point renders."""
        cams, ray_cam = self._bundle_cameras(ray_bundle)
        rnd = self._fused_renderer()
        rnd.opts.jitter = float(self.neural_points.jitter)
        rnd.opts.seed = self._render_calls & 0xFFFFFFFF
        self._render_calls += 1
        dirs = ray_bundle.directions.to(self._device)
        out = rnd.render_views(dirs, cams, dirs.reshape(-1, 3).shape[0], ray_cam=ray_cam)
        return {"coarse_raycolor": out["rgb"], "ray_mask": out["ray_mask"], "depth": out["depth"],
                "accumulation": out["acc"]}

    def _get_outputs_fused_train(self, ray_bundle):
        """Training step on the fused path: pnr_render forwards (no clamp, the reference's 0.3 jitter with a fresh
        seed per call), pnr_render_backward behind a torch.autograd.Function for d loss / d {points_embeding,
        points_color, points_dir, MLP weights} -- what autograd derives for studio_model.py:263-399.
        `conf_coefficient` (studio_model.py:288-292) is gathered here with torch ops so that its loss term reaches
        points_conf: the reference's tensor is [1,R'',SR,K] with unfilled slots reading point 0
        (studio_utils.py:193-199, clamp(pidx, 0)); the same multiset of values is returned flat, which is all the
        loss (a mean) looks at."""
        cams, ray_cam = self._bundle_cameras(ray_bundle)
        rnd = self._fused_renderer(train=True)
        rnd.opts.jitter = float(self.neural_points.jitter)
        rnd.opts.seed = self._render_calls & 0xFFFFFFFF
        self._render_calls += 1
        npts = self.neural_points
        mlp = []
        for name in MLP_TENSOR_ORDER:
            mod = self.get_submodule(name)
            mlp += [mod.weight, mod.bias]
        rgb, ray_mask = _FusedRenderFn.apply(rnd, ray_bundle.directions.to(self._device), cams, ray_cam,
                                             npts.points_embeding, npts.points_color, npts.points_dir, *mlp)
        cnt = rnd.last_counters
        R = ray_bundle.directions.reshape(-1, 3).shape[0]
        pidx = rnd.taps(R)["smp_pidx"][:cnt["samples_selected"]].reshape(-1).long()
        conf = npts.points_conf[0, :, 0]
        cv = conf[pidx[pidx >= 0]]
        n_slots = cnt["rays_kept"] * self.config.SR * self.config.K
        conf_all = torch.cat([cv, conf[0:1].expand(max(n_slots - cv.numel(), 0))])
        conf_coefficient = conf_all - (conf_all - torch.clamp(conf_all, min=0.0001, max=1)).detach()
        return {"coarse_raycolor": rgb, "ray_mask": ray_mask, "conf_coefficient": conf_coefficient}

    # ---- point growing / pruning (absent from the reference's plugin, present in its
    # legacy trainer: run/train_studio.py:335-444,676-735) -------------------------------------------------------------
    @torch.no_grad()
    def get_probe_outputs(self, ray_bundle) -> Dict[str, torch.Tensor]:
        """One eval render + the probing outputs of the legacy model (`opt.prob == 1`,
        models/neural_points_volumetric_model.py:331-352) under the legacy key names: coarse_raycolor, ray_mask,
        ray_max_shading_opacity, ray_max_sample_loc_w, ray_max_far_dist, shading_avg_{color,dir,conf,embedding} --
        what probe_hole (run/train_studio.py:335-429) scatters into its per-pixel maps."""
        if not self._fusable():
            raise RuntimeError("get_probe_outputs needs the fused HIP path (default network shape)")
        was_training = self.training
        self.eval()
        try:
            out = self._get_outputs_fused(ray_bundle)
            out.update(self._renderer.probe())
        finally:
            self.train(was_training)
        return out

    def prune_points(self, thresh: float) -> int:
        """models/neural_points_volumetric_model.py prune_points -> neural_points.prune (neural_points.py:341-364).  The
        caller re-creates its optimisers afterwards (the parameters are new tensors), as the reference's trainer does."""
        return self.neural_points.prune(thresh)

    def grow_points(self, add_xyz, add_embedding, add_color, add_dir, add_conf) -> int:
        """neural_points.py:367-393 (called at run/train_studio.py:714)."""
        return self.neural_points.grow_points(add_xyz, add_embedding, add_color, add_dir, add_conf)

    def get_outputs(self, ray_bundle):
        if self.mlp_base is None:
            raise ValueError("populate_fields() must be called before get_outputs")
        if not self.training and not torch.is_grad_enabled() and self._fusable():
            return self._get_outputs_fused(ray_bundle)
        if self.training and torch.is_grad_enabled() and self._fusable() and \
                getattr(self.config, "hip_fused_training", True):
            return self._get_outputs_fused_train(ray_bundle)
        return self._get_outputs_autograd(ray_bundle)

    # ---- the reference's op sequence (training) -----------------------------------------------------------
    def _get_outputs_autograd(self, ray_bundle):
        """studio_model.py:263-399 on device tensors; the query inside neural_points() is the HIP op."""
        (sampled_color, sampled_Rw2c, sampled_dir, sampled_embedding, sampled_xyz_pers, sampled_xyz, sampled_conf,
         sample_loc_tensor, sample_loc_w_tensor, sample_pnt_mask, sample_ray_dirs_tensor, vsize_np,
         ray_mask_tensor) = self.neural_points(ray_bundle)
        dev = sample_loc_w_tensor.device
        sample_valid = torch.any(sample_pnt_mask, dim=-1).view(-1)
        total_len = len(sample_valid)
        in_shape = sample_loc_w_tensor.shape
        B, R, SR, K = sample_pnt_mask.shape
        if R > 0:
            xdist = sampled_xyz_pers[..., 0] * sampled_xyz_pers[..., 2] - (sample_loc_tensor[..., 0] * sample_loc_tensor[..., 2])[..., None]
            ydist = sampled_xyz_pers[..., 1] * sampled_xyz_pers[..., 2] - (sample_loc_tensor[..., 1] * sample_loc_tensor[..., 2])[..., None]
            zdist = sampled_xyz_pers[..., 2] - sample_loc_tensor[..., 2][..., None]
            dists = torch.cat([sampled_xyz - sample_loc_w_tensor[..., None, :], torch.stack([xdist, ydist, zdist], -1)], -1)
        else:
            dists = torch.zeros([B, R, SR, K, 6], device=dev)
        axis_weight = torch.as_tensor(self.config.axis_weight, dtype=torch.float32, device=dev)[None, None, None, None, :]
        weight = self.linear(dists, sample_pnt_mask, axis_weight=axis_weight)
        weight = weight / torch.clamp(torch.sum(weight, dim=-1, keepdim=True), min=1e-8)
        conf_coefficient = None
        if self.training:
            conf = sampled_conf[..., 0]
            conf_coefficient = conf - (conf - torch.clamp(conf, min=0.0001, max=1)).detach()

        flat = sample_pnt_mask.view(-1)
        Rt = sampled_Rw2c.transpose(-1, -2)
        viewdirs = self.direction_encoding(sample_ray_dirs_tensor.reshape(-1, 3) @ Rt)
        ori_viewdirs, viewdirs = viewdirs[..., :3], viewdirs[..., 3:]
        viewdirs = viewdirs[sample_valid, :]
        d = dists.view(-1, 6)[flat, :]
        d = torch.cat([d[..., :3] @ Rt, d[..., 3:]], dim=-1)
        feat = sampled_embedding.reshape(-1, sampled_embedding.shape[-1])[flat, :]
        feat = torch.cat([feat, self.feature_encoding(feat), self.dists_encoding(d)], dim=-1)
        weight = weight.view(B * R * SR, K, 1)
        feat = self.mlp_base(feat)
        col = sampled_color.reshape(-1, 3)[flat, :]
        sdir = sampled_dir.reshape(-1, 3)[flat, :] @ Rt
        ov = ori_viewdirs[..., None, :].repeat(1, K, 1).view(-1, 3)[flat, :]
        feat = torch.cat([feat, col, sdir - ov, torch.sum(sdir * ov, dim=-1, keepdim=True)], dim=-1)
        feat = self.mlp_head(feat)
        alpha = self.field_output_density(feat)
        holder = torch.zeros([B * R * SR * K, 1], dtype=torch.float32, device=dev)
        holder[flat, :] = alpha
        alpha = torch.sum(holder.view(B * R * SR, K, 1) * weight, dim=-2).view(-1, 1)[sample_valid, :]
        holder = torch.zeros([B * R * SR * K, feat.shape[-1]], dtype=torch.float32, device=dev)
        holder[flat, :] = feat
        feat = torch.sum(holder.view(B * R * SR, K, -1) * weight, dim=-2).view(-1, feat.shape[-1])[sample_valid, :]
        color = self.field_output_color(self.mlp_color(torch.cat([feat, viewdirs], dim=-1)))
        color = color * (1 + 2 * 0.001) - 0.001
        decoded = torch.zeros([total_len, 4], dtype=torch.float32, device=dev)
        decoded[sample_valid] = torch.cat([alpha, color], dim=-1)
        decoded = decoded.view(in_shape[:-1] + (4,))
        sample_valid = sample_valid.view(in_shape[:-1])

        ray_dist = torch.cummax(sample_loc_tensor[..., 2], dim=-1)[0]
        ray_dist = torch.cat([ray_dist[..., 1:] - ray_dist[..., :-1],
                              torch.full((B, R, 1), vsize_np[2], device=dev)], dim=-1)
        m = torch.logical_or(ray_dist < 1e-8, ray_dist > 2 * vsize_np[2]).to(torch.float32)
        ray_dist = (ray_dist * (1.0 - m) + m * vsize_np[2]) * sample_valid.float()
        sigma = decoded[..., 0] * sample_valid.float()
        opacity = 1 - torch.exp(-sigma * ray_dist)
        acc_t = torch.cumprod(1. - opacity + 1e-10, dim=-1)
        acc_t = torch.cat([torch.ones((B, R, 1), device=dev), acc_t[:, :, :-1]], dim=-1)
        blend_weight = (opacity * acc_t).unsqueeze(-1)
        output = {"coarse_raycolor": self.rgb_renderer(rgb=decoded[..., 1:4], weights=blend_weight),
                  "ray_mask": ray_mask_tensor}
        output = self.fill_invalid(output)
        output["ray_mask"] = output["ray_mask"].squeeze(0)
        if self.training:
            output["conf_coefficient"] = conf_coefficient
        return output

    # ---- the rest of the plugin surface ---------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        """studio_model.py:401-413: `neural_points` = parameters named neural_points.points*, `fields` = the rest."""
        if self.mlp_base is None:
            raise ValueError("populate_fields() must be called before get_param_groups")
        named = list(self.named_parameters())
        return {"neural_points": [p for n, p in named if n.startswith("neural_points.points")],
                "fields": [p for n, p in named if not n.startswith("neural_points.points")]}

    def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
        """The reference inherits nerfstudio's empty list.  Here one callback runs after every optimiser step:
        point features and MLP weights changed, so the PACKED copies (point rows, MFMA-ordered weights) are marked
        stale and re-packed lazily by the next render.  The voxel structure, the SceneHIP / RendererHIP objects and
        their workspaces stay: points_xyz is frozen (studio_utils.py:84), and a real change of it is caught by the
        (data_ptr, _version) key of NeuralPoints.fused_scene."""
        def _invalidate(step: int = 0):
            self.neural_points.invalidate_packed()
            self._weights_key = None
        return [TrainingCallback(where_to_run=[TrainingCallbackLocation.AFTER_TRAIN_ITERATION], func=_invalidate)]

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """studio_model.py:415-431."""
        device = outputs["coarse_raycolor"].device
        image = batch["image"].to(device)
        keep = (outputs["ray_mask"] > 0)[..., None].expand(-1, 3)
        masked_output = torch.masked_select(outputs["coarse_raycolor"], keep).reshape(-1, 3)
        masked_gt = torch.masked_select(image, keep).reshape(-1, 3)
        loss_dict = {"ray_masked_coarse_raycolor_loss": self.mask_loss(masked_gt, masked_output) + 1e-6}
        if self.training:
            val = torch.clamp(outputs["conf_coefficient"], self.config.zero_epsilon, 1 - self.config.zero_epsilon)
            loss_dict["conf_coefficient_loss"] = \
                torch.mean(torch.log(val) + torch.log(1 - val)) * self.config.zero_one_loss_weights
        coeff = getattr(self.config, "loss_coefficients", None) or {}
        return {k: v * coeff.get(k, 1.0) for k, v in loss_dict.items()}

    def get_image_metrics_and_images(self, outputs: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]
                                     ) -> Tuple[Dict[str, float], Dict[str, torch.Tensor]]:
        """studio_model.py:433-464, same metric and image keys.  The reference reshapes to a hard-coded 800 x 800
        (:437); here the extent comes from the batch image, so other resolutions work."""
        image = batch["image"].to(outputs["coarse_raycolor"].device)
        H, W = image.shape[0], image.shape[1]
        outputs["ray_masked_coarse_raycolor"] = outputs["coarse_raycolor"].reshape(H, W, 3)
        rgb = outputs["ray_masked_coarse_raycolor"]
        combined_rgb = torch.cat([image, rgb], dim=1)
        # [H, W, C] -> [1, C, H, W] for the metrics
        image = torch.moveaxis(image, -1, 0)[None, ...]
        rgb = torch.moveaxis(rgb, -1, 0)[None, ...]
        metrics_dict = {
            "psnr": float(self.psnr(image, rgb)),
            "skimage_ssim": float(self.skimage_ssim(image, rgb)),
            "torchmetrics_ssim": float(self.torchmetrics_ssim(image, rgb)),
            "lpips": float(self.lpips(image, rgb)),
            "lpips_vgg": float(self.lpips_vgg(image, rgb)),
            "rmse": float(self.skimage_rmse(image, rgb)),
        }
        return metrics_dict, {"img": combined_rgb}

    def linear(self, dists, pnt_mask, axis_weight=None):
        """studio_model.py:467-475."""
        if axis_weight is None or (axis_weight[..., 0] == 1 and axis_weight[..., 2] == 1):
            weights = 1. / torch.clamp(torch.norm(dists[..., :3], dim=-1), min=1e-6)
        else:
            weights = 1. / torch.clamp(
                torch.sqrt(torch.sum(torch.square(dists[..., :2]), dim=-1)) * axis_weight[..., 0] +
                torch.abs(dists[..., 2]) * axis_weight[..., 1], min=1e-6)
        return pnt_mask * weights

    def fill_invalid(self, output):
        """studio_model.py:491-504, without the torch.nonzero host sync."""
        ray_mask = output["ray_mask"]
        B, OR = ray_mask.shape
        rgb = output["coarse_raycolor"]
        full = torch.ones([B, OR, 3], dtype=rgb.dtype, device=rgb.device) * self._background_color.to(rgb.device)
        full[ray_mask > 0] = rgb.reshape(-1, 3)
        output["coarse_raycolor"] = full.squeeze(0)
        return output
