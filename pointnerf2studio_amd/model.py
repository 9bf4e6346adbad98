"""Host-side mirror of the reference's nerfstudio Model (pointnerf/nerfstudio/studio_model.py):
`PointNerfConfig` with the same fields and defaults, `PointNerf` with the same module names (so state
dicts and optimiser groups carry over), `get_outputs` / `get_param_groups` / `get_loss_dict` /
`get_training_callbacks` / `fill_invalid` / `linear`.

Every body of get_outputs is the fused HIP path (include/pnr.h):
  * without autograd (the path the metric times): `get_outputs_for_camera_ray_bundle` renders the whole [H, W] camera
    bundle in ONE pnr_render_views call (nerfstudio's inherited loop would make 278 calls of 2304 rays per 800 x 800
    image, studio_config.py:25); `get_outputs` renders a ray bundle in one call;
  * with autograd (training steps, and eval-mode calls with gradients enabled): pnr_render_views forwards and
    pnr_render_backward backwards behind one torch.autograd.Function; the point gradients are accumulated straight
    into persistent dense buffers handed out as `.grad` (no 768-MB zero fill per step), the packed point rows are
    refreshed by the render itself from the bound parameters (no O(N) re-pack per step), and a step contains ONE
    device-to-host read (the bundle's camera, near and far).
There is no PyTorch-op render in this package and no CPU path anywhere: a configuration the fused kernels do not cover
(or hip_fused_training=False) raises.  The reference's op sequence under torch autograd, which the fused training step is
compared with, is test infrastructure (tests/autograd_reference_path.py) and reaches the model only through the
`unfused_outputs_fn` hook a test installs.
"""
from __future__ import annotations

import dataclasses
import glob
import os
import weakref
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
from .optim import publish_rows
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
    # get_outputs_for_camera_ray_bundle: the whole camera bundle in ONE fused call (False = nerfstudio's chunk loop over
    # eval_num_rays_per_chunk rays, kept for the equality test)
    hip_eval_one_call: bool = True
    # training: point gradients accumulated by the backward kernels straight into persistent dense buffers that are
    # handed out as `.grad` and cleaned row-wise (False = a fresh zero-filled dense tensor per step through autograd)
    hip_sparse_point_grads: bool = True
    # training: a render workspace sized for the worst case (every ray selects SR samples) cannot overflow, so a step
    # needs no host read of the counters; used while render + backward workspaces stay below this many GiB
    hip_train_workspace_gb: float = 48.0
    # renders that a backward follows write the backward's activation tape themselves (pnr_render_opts_t.d_tape): the
    # backward skips its four recompute GEMMs
    hip_tape_from_render: bool = True
    # training: the confidence regulariser (studio_model.py:288-292,427-429) as a kernel pair over the render's neighbour
    # lists; the outputs then carry `conf_coefficient_loss_term` (the mean get_loss_dict weights) instead of the
    # `conf_coefficient` tensor.  False: `conf_coefficient` values + `conf_coefficient_weights` multiplicities (torch ops)
    hip_conf_loss_kernel: bool = True
    # every bundle handed to the model comes from ONE camera (the reference's datamanager picks one image per batch,
    # studio_datamanager.py:62-81, and the reference itself reads origins[0] / camrotc2w[0] only, studio_utils.py:148-152):
    # the pose is then read by the kernels from the bundle's device tensors (pnr_render_pose) and, with a collider that
    # states its planes, a call issues no device-to-host read at all.  False: bundles may mix cameras (one read to know)
    hip_single_camera_bundles: bool = False

    def __post_init__(self):
        if self.path_point_cloud is not None:
            if not Path(self.path_point_cloud).exists():
                raise RuntimeError(f"PointCloud path {self.path_point_cloud} does not exist")


class _FusedRenderFn(torch.autograd.Function):
    """pnr_render_views forwards, pnr_render_backward backwards (include/pnr.h).  Inputs after `cap`: points_embeding,
    points_color, points_dir and the nine (weight, bias) pairs in MLP_TENSOR_ORDER.  cap: capacity of the render
    workspace in selected samples -- the worst case R * SR (no overflow possible, no host read) or None (grown on
    demand behind a read of the counters)."""

    @staticmethod
    def forward(ctx, model, rnd, dirs, cams, ray_cam, cap, emb, color, pdir, *mlp):
        R = dirs.shape[0]
        if isinstance(cams, tuple):     # (campos, camrotc2w on the device, near, far): _device_pose
            out = rnd.render_pose(dirs, cams[0], cams[1], cams[2], cams[3], cap_samples=cap, sync_counters=cap is None)
        else:
            out = rnd.render_views(dirs, cams, R, ray_cam=ray_cam, cap_samples=cap, sync_counters=cap is None)
        ctx.model, ctx.rnd = model, rnd
        ctx.shapes = (emb.shape, color.shape, pdir.shape)
        ctx.state = {name + suf: mlp[2 * i + j] for i, name in enumerate(MLP_TENSOR_ORDER)
                     for j, suf in enumerate((".weight", ".bias"))}
        ctx.call = rnd.calls
        ctx.mark_non_differentiable(out["ray_mask"])
        return out["rgb"], out["ray_mask"]

    @staticmethod
    def backward(ctx, g_rgb, _g_mask):
        rnd, model = ctx.rnd, ctx.model
        if rnd.calls != ctx.call:
            raise RuntimeError("fused training: the renderer ran another render before backward(); its workspace "
                               "no longer holds this step's sample lists")
        es, cs, ds = ctx.shapes
        targets = model._point_grad_targets() if model.config.hip_sparse_point_grads else None
        if targets is not None:
            # the kernels add the touched rows into the tensors that ARE (or become) the parameters' .grad: autograd
            # is told "no gradient" for the three point tensors
            g = rnd.backward(g_rgb, ctx.state, es[-2], into=targets)
            model._after_point_backward(rnd, targets)
            grads = [None, None, None]
        else:
            g = rnd.backward(g_rgb, ctx.state, es[-2])
            grads = [g["embedding"].view(es), g["color"].view(cs), g["dir"].view(ds)]
        for name in MLP_TENSOR_ORDER:
            grads += [g[name + ".weight"], g[name + ".bias"]]
        return (None, None, None, None, None, None, *grads)


class _MaskedMSEFn(torch.autograd.Function):
    """MSELoss over the kept rays (studio_model.py:419-424) as a masked sum: mean((image - rgb)^2 over kept rays) + 1e-6.
    The backward is written out (two small kernels) instead of autograd's dozen for the same expression."""

    @staticmethod
    def forward(ctx, rgb, image, ray_mask):
        keep = (ray_mask > 0)[..., None].to(rgb.dtype)
        diff = (image - rgb) * keep
        n = 3.0 * torch.sum(keep)
        ctx.save_for_backward(diff, n)
        return torch.sum(diff * diff) / n + 1e-6

    @staticmethod
    def backward(ctx, g):
        diff, n = ctx.saved_tensors
        return diff * (-2.0 * g / n), None, None


class _ConfLossFn(torch.autograd.Function):
    """mean(log v + log(1 - v)) over the reference's conf_coefficient tensor of the last render (pnr_conf_loss) and its
    gradient w.r.t. points_conf (pnr_conf_loss_backward): studio_model.py:288-292,427-429 without the [1,R'',SR,K] gather."""

    @staticmethod
    def forward(ctx, rnd, eps, conf):
        out = rnd.conf_loss(conf.detach(), eps)
        ctx.rnd, ctx.eps, ctx.call = rnd, eps, rnd.calls
        ctx.save_for_backward(conf, out)
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g, _g_out):
        conf, out = ctx.saved_tensors
        if ctx.rnd.calls != ctx.call:
            raise RuntimeError("fused training: the renderer ran another render before backward(); its workspace no longer "
                               "holds this step's neighbour lists")
        grad = torch.zeros_like(conf, memory_format=torch.contiguous_format)
        ctx.rnd.conf_loss_backward(conf.detach(), ctx.eps, out, g, grad)
        return None, None, grad


class PointNerf(Model):
    """studio_model.py:121-505."""
    config: PointNerfConfig
    # hook: (model, ray_bundle) -> outputs for calls the fused path does not serve.  None in the product; the tests that
    # compare the fused training step with torch autograd install tests/autograd_reference_path.get_outputs_autograd
    unfused_outputs_fn = None
    _ALL_ROWS = "all rows"              # marker in _gdirty: clear the whole buffer instead of listed rows
    _MAX_PENDING_ROW_LISTS = 4

    def __init__(self, config: PointNerfConfig, cameras=None, point_state_dict: Optional[Dict] = None,
                 **kwargs) -> None:
        self._point_state_dict = point_state_dict   # lets tests / benches hand the cloud over in memory
        super().__init__(config=config, **kwargs)
        self._point_initialized = False
        self.cameras = cameras
        self._device = "cuda"
        self._renderers: Dict[bool, RendererHIP] = {}     # by eval_clamp
        self._weights: Optional[WeightsHIP] = None
        self._weights_key = None
        self._render_calls = 0
        self._cam_memo = None       # (weakref of the last bundle, versions, cameras): see _bundle_cameras
        self._gbuf: Dict[str, torch.Tensor] = {}          # persistent dense point-gradient buffers, by tensor
        self._gdirty: Dict[str, list] = {}                # rows written into them since they were last all-zero
        self.grad_exchange = None   # distributed.GradExchange when the training step is data-parallel
        self.host_reads = 0         # device-to-host reads issued by the fused paths (tests hold the count per step)
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

    def _fused_renderer(self, clamp: bool = True, live: bool = False) -> RendererHIP:
        """The renderer for eval_clamp = `clamp` (nerfstudio's RGBRenderer clamps outside training only); each owns its
        workspace.  live: the parameter tensors are bound to the scene (training steps, see NeuralPoints.fused_scene)."""
        scene = self.neural_points.fused_scene(live=live)
        sd = self._mlp_state()
        rkey = (self.neural_points.points_Rw2c.data_ptr(), self.neural_points.points_Rw2c._version)
        key = tuple((t.data_ptr(), t._version) for t in sd.values()) + rkey
        if self._weights is None or key != self._weights_key:
            dev = self.neural_points.points_xyz.device
            if self._weights is None:
                self._weights = WeightsHIP()
                self._weights_rkey = None
            if self._weights_rkey == rkey and live:
                # an optimiser step moved the layers, Rw2c is what it was: one launch, no synchronisation, only the
                # forms this arithmetic mode reads (the other mode's are rebuilt by the next full pack)
                self._weights.update(sd, dev, getattr(self.config, "hip_mlp_mode", "fp32"))
                self._weights_partial = True
            else:
                self._weights.pack(sd, self.neural_points.points_Rw2c.detach(), dev)
                self._weights_rkey, self._weights_partial = rkey, False
            self._weights_key = key
        elif getattr(self, "_weights_partial", False) and not live:
            # (eval after training steps: bring every form up to date once)
            self._weights.pack(sd, self.neural_points.points_Rw2c.detach(), self.neural_points.points_xyz.device)
            self._weights_partial = False
        c = self.config
        rnd = self._renderers.get(clamp)
        if rnd is None or rnd.scene is not scene:
            rnd = RendererHIP(scene, self._weights, SR=c.SR, K=c.K, D=c.z_depth_dim,
                              radius_limit=float(self.neural_points.radius_limit_np), vsize_z=c.vsize[2],
                              eval_clamp=clamp, bg=self._background_color.tolist(),
                              precision=getattr(c, "hip_mlp_mode", "fp32"))
            self._renderers[clamp] = rnd
        rnd.mlp_state = sd
        return rnd

    # (kept names: the two renderers the tests and tools look at)
    @property
    def _renderer(self) -> Optional[RendererHIP]:
        return self._renderers.get(True)

    @property
    def _renderer_train(self) -> Optional[RendererHIP]:
        return self._renderers.get(False)

    def forward(self, ray_bundle):
        """nerfstudio's Model.forward (collider, then get_outputs) [ns-mem]; it also remembers WHICH bundle just got its
        nears / fars from the collider, so that _bundle_cameras may take the planes from the collider's memo instead of
        reading them back from the device."""
        collider = getattr(self, "collider", None)
        if collider is not None:
            ray_bundle = collider(ray_bundle)
            self._note_collided(ray_bundle)
        return self.get_outputs(ray_bundle)

    def _note_collided(self, ray_bundle) -> None:
        try:
            self._collided = weakref.ref(ray_bundle)
        except TypeError:
            self._collided = None

    def _collider_planes(self, ray_bundle):
        """(key, (near, far) or None): a NearFarCollider writes ones * plane, a function of its attributes and mode only
        -- once the planes of such a collider state have been read they are known for every later bundle it fills.
        Other colliders (no near_plane / far_plane attributes) and bundles that did not come through forward(): no key."""
        collider = getattr(self, "collider", None)
        ref = getattr(self, "_collided", None)
        if collider is None or ref is None or ref() is not ray_bundle or not hasattr(collider, "near_plane") \
                or not hasattr(collider, "far_plane"):
            return None, None
        key = (id(collider), bool(getattr(collider, "training", False)), float(collider.near_plane),
               float(collider.far_plane), bool(getattr(collider, "reset_near_plane", True)))
        return key, getattr(self, "_plane_memo", {}).get(key)

    def _bundle_cameras(self, ray_bundle, one_camera: bool = False, owner=None):
        """The cameras of a bundle.  The reference assumes ONE per bundle and reads origins[0] / camrotc2w[0] / nears[0]
        / fars[0] with three host reads (studio_utils.py:148-155); here everything the host needs -- position, rotation,
        near, far and an "all rays share them" flag computed on the device -- arrives in ONE 15-float read; a bundle
        OBJECT seen before (`owner`, default the bundle itself: same tensors, same versions) costs none, and the planes
        a NearFarCollider wrote are read once per collider state.  one_camera: the caller vouches for a single camera
        (a camera ray bundle: nerfstudio generates it from one camera), the flag is not even computed.
        Bundles that mix cameras -- nerfstudio's random-pixel batches over several images
        -- are rendered in ONE pnr_render_views call with a per-ray camera index, up to PNR_MAX_CAMS cameras.
        Returns (cams [(pos[3], rot[9], near, far)] as host floats, ray_cam int32 [R] or None)."""
        meta = ray_bundle.metadata["camrotc2w"]
        plane_key, planes = self._collider_planes(ray_bundle)
        owner = ray_bundle if owner is None else owner
        tensors = (ray_bundle.origins, meta) + (() if plane_key is not None else (ray_bundle.nears, ray_bundle.fars))
        versions = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors) + (one_camera, plane_key)
        memo = self._cam_memo
        if memo is not None and memo[0]() is owner and memo[1] == versions and (plane_key is None or planes is not None):
            cams, ray_cam = memo[2]
            if planes is not None:
                cams = [(c[0], c[1], planes[0], planes[1]) for c in cams]
            return cams, ray_cam
        dev = self._device
        o = ray_bundle.origins.reshape(-1, 3)
        per_ray = tuple(meta.shape) != (3, 3)
        m = meta.reshape(-1, 9) if per_ray else meta.reshape(1, 9)
        mixed_possible = per_ray and not one_camera and o.shape[0] > 1
        same = ((o == o[0]).all() & (m == m[0]).all()).reshape(1) if mixed_possible else o.new_ones(1, dtype=torch.bool)
        head = torch.cat([o[0].reshape(3).to(dev, torch.float32), m[0].reshape(9).to(dev, torch.float32),
                          ray_bundle.nears.reshape(-1)[:1].to(dev, torch.float32),
                          ray_bundle.fars.reshape(-1)[:1].to(dev, torch.float32),
                          same.to(dev, torch.float32)]).cpu().tolist()      # the ONE host read
        self.host_reads += 1
        near, far = head[12], head[13]
        if plane_key is not None:
            if not hasattr(self, "_plane_memo"):
                self._plane_memo = {}
            self._plane_memo[plane_key] = (near, far)
        if head[14] != 0.0:
            result = ([(head[0:3], head[3:12], near, far)], None)
        else:
            key = torch.cat([o, m], dim=1).to(dev)
            uniq, inv = torch.unique(key, dim=0, return_inverse=True)
            if uniq.shape[0] > MAX_CAMS:
                raise RuntimeError(f"a ray bundle may mix at most {MAX_CAMS} cameras, got {uniq.shape[0]}")
            uniq = uniq.cpu().tolist()
            self.host_reads += 1
            result = ([(u[:3], u[3:], near, far) for u in uniq], inv.to(torch.int32))
        try:
            self._cam_memo = (weakref.ref(owner), versions, result)
        except TypeError:      # a bundle type that cannot be weakly referenced: no memo
            self._cam_memo = None
        return result

    def _device_pose(self, ray_bundle):
        """(campos [3], camrotc2w [9] on the device, near, far) when the bundle can be rendered without a host read:
        hip_single_camera_bundles vouches for one camera and the planes are known from the collider's memo; else None
        (the caller takes _bundle_cameras, whose read also fills that memo)."""
        if not getattr(self.config, "hip_single_camera_bundles", False):
            return None
        plane_key, planes = self._collider_planes(ray_bundle)
        if planes is None:
            return None
        meta = ray_bundle.metadata["camrotc2w"]
        rot = meta.reshape(9) if tuple(meta.shape) == (3, 3) else meta.reshape(-1, 9)[0]
        pos = ray_bundle.origins.reshape(-1, 3)[0]
        dev = self._device
        return (pos.to(dev, torch.float32).contiguous(), rot.to(dev, torch.float32).contiguous(), planes[0], planes[1])

    def _next_seed(self) -> int:
        seed = self._render_calls & 0xFFFFFFFF
        self._render_calls += 1
        return seed

    def _get_outputs_fused(self, ray_bundle, one_camera: bool = False, owner=None):
        """No autograd: one pnr_render_views call.  Jitter: the reference draws torch.rand jitter even at eval
        (studio_utils.py:166, hard-coded 0.3); the fused path uses the same fraction (`neural_points.jitter`) with the
        library's counter-based uniforms and a fresh seed per call.  Set `neural_points.jitter = 0` for deterministic
        mid-point renders.  The clamp follows the module's mode as nerfstudio's RGBRenderer does."""
        pose = self._device_pose(ray_bundle)
        if pose is None:
            cams, ray_cam = self._bundle_cameras(ray_bundle, one_camera, owner)
        rnd = self._fused_renderer(clamp=not self.training)
        rnd.opts.jitter = float(self.neural_points.jitter)
        rnd.opts.seed = self._next_seed()
        rnd.opts.early_stop_eps = 0.0 if self.training else float(getattr(self.config, "hip_early_stop_eps", 0.0))
        rnd.tape = False
        dirs = ray_bundle.directions.to(self._device).reshape(-1, 3)
        # a small bundle (eval batches, chunks) gets a workspace that cannot overflow: no counters to read back; a whole
        # frame keeps the capacity earlier frames needed and checks the overflow counter at the end of the call
        cap = self._worst_case_cap(rnd, dirs.shape[0], backward=False)
        if pose is not None:
            out = rnd.render_pose(dirs, pose[0], pose[1], pose[2], pose[3], cap_samples=cap, sync_counters=cap is None)
        else:
            out = rnd.render_views(dirs, cams, dirs.shape[0], ray_cam=ray_cam, cap_samples=cap, sync_counters=cap is None)
        if cap is None:
            self.host_reads += 1
        return {"coarse_raycolor": out["rgb"], "ray_mask": out["ray_mask"], "depth": out["depth"],
                "accumulation": out["acc"]}

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle) -> Dict[str, torch.Tensor]:
        """The whole [H, W] camera bundle (studio_datamanager.py:104-110) in ONE fused call.  nerfstudio's inherited
        method cuts it into eval_num_rays_per_chunk = 2304 rays (studio_config.py:25) and calls forward per chunk: 278
        calls per 800 x 800 image, each of which -- in the reference -- rebuilds the voxel grid.  Same outputs, viewed
        as [H, W, -1]; at jitter 0 bit-identical to the chunk loop (rays are independent), with jitter the one call
        draws one seed for the image instead of one per chunk."""
        if not (self._fusable() and getattr(self.config, "hip_eval_one_call", True)):
            return super().get_outputs_for_camera_ray_bundle(camera_ray_bundle)
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        num_rays = image_height * image_width
        ray_bundle = camera_ray_bundle.get_row_major_sliced_ray_bundle(0, num_rays)     # flat views, no copy
        if getattr(self, "collider", None) is not None:
            ray_bundle = self.collider(ray_bundle)
            self._note_collided(ray_bundle)
        out = self._get_outputs_fused(ray_bundle, one_camera=True, owner=camera_ray_bundle)
        return {k: v.view(image_height, image_width, -1) for k, v in out.items() if torch.is_tensor(v)}

    # ---- autograd on the fused path -----------------------------------------------------------------------
    def _worst_case_cap(self, rnd: RendererHIP, R: int, backward: bool) -> Optional[int]:
        """Worst-case capacity R * SR when the render (+ backward) workspaces of that size fit hip_train_workspace_gb:
        the render cannot overflow then and nothing has to be read back.  None: grow on demand (one read per call)."""
        worst = R * self.config.SR
        memo = getattr(self, "_cap_memo", {})
        key = (id(rnd.scene), R, backward)
        if key not in memo:
            need = rnd.lib.pnr_render_workspace_bytes_for(rnd.scene.handle, rnd.opts, R, worst)
            if backward:
                need += rnd.lib.pnr_backward_workspace_bytes(worst, self.config.K)
            memo[key] = worst if need <= float(getattr(self.config, "hip_train_workspace_gb", 48.0)) * 2 ** 30 else None
            self._cap_memo = memo
        return memo[key]

    def _point_grad_targets(self) -> Dict[str, Optional[torch.Tensor]]:
        """Where pnr_render_backward accumulates the point gradients: for every trainable point tensor its existing
        `.grad` if it has one, otherwise this model's persistent zero buffer, which BECOMES `.grad`.  The buffer is kept
        all-zero outside the rows written since it was last handed out; those rows are cleared (pnr_point_grads_clear,
        O(rows)) when a `zero_grad(set_to_none=True)` has detached it -- torch's autograd would allocate and fill a dense
        768-MB gradient per step for the reference's index_select (studio_utils.py:199-207)."""
        npts = self.neural_points
        N = npts.points_xyz.shape[0]
        rnd = self._renderers.get(not self.training) or next(iter(self._renderers.values()))
        out: Dict[str, Optional[torch.Tensor]] = {}
        clear: Dict[int, list] = {}   # id(row list) -> [index, count, {key: buffer}]: one launch per list for all tensors
        for key, p in (("embedding", npts.points_embeding), ("color", npts.points_color), ("dir", npts.points_dir)):
            if not p.requires_grad:
                out[key] = None
                continue
            buf = self._gbuf.get(key)
            if buf is None or buf.shape != p.shape or buf.device != p.device:
                buf = self._gbuf[key] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                self._gdirty[key] = []
            if p.grad is None:
                if self._gdirty[key] is self._ALL_ROWS:
                    buf.zero_()     # (see _after_point_backward: too many row lists were pending)
                else:
                    for index, count in self._gdirty[key]:
                        clear.setdefault(id(index), [index, count, {}])[2][key] = buf
                self._gdirty[key] = []
                p.grad = buf
            elif self.grad_exchange is not None and self.grad_exchange.world > 1:
                # the row exchange sends what the touched rows of .grad hold AFTER the local accumulation: with an earlier
                # backward's sums still in there, rows already exchanged would be sent again (wrong scale, ranks diverge)
                raise RuntimeError(
                    f"points_{key}.grad is set before a data-parallel backward: gradient accumulation over several "
                    "backward calls (and no_sync) is not supported with the sparse row exchange -- call "
                    "zero_grad(set_to_none=True) between steps")
            elif p.grad.data_ptr() != buf.data_ptr() and not (p.grad.is_contiguous() and p.grad.dtype == torch.float32):
                raise RuntimeError(f"points_{key}.grad is not a contiguous float32 tensor")
            out[key] = p.grad
        for index, count, bufs in clear.values():
            rnd.clear_point_grads(bufs.get("embedding"), bufs.get("color"), bufs.get("dir"), N, index, count)
        return out

    def _after_point_backward(self, rnd: RendererHIP, targets) -> None:
        """Book-keeping behind a backward that accumulated into the dense buffers: the rows it touched (device list +
        device count, no host read) are remembered for the next clean-up; a data-parallel step exchanges them here."""
        index, count = rnd.touched()
        if self.grad_exchange is not None and self.grad_exchange.world > 1:
            index, count = self.grad_exchange.exchange_dense_rows(index, count, targets)
            self.host_reads += 1
        # the optimiser half of the step: PointRowAdam (optim.py) moves only rows that ever had a gradient -- these ones
        npts = self.neural_points
        publish_rows((npts.points_embeding, npts.points_color, npts.points_dir, npts.points_conf), index, count)
        for key, t in targets.items():
            buf = self._gbuf.get(key)
            if t is not None and buf is not None and t.data_ptr() == buf.data_ptr():
                pending = self._gdirty[key]
                if pending is self._ALL_ROWS:
                    continue
                # the caller keeps `.grad` set (zero_grad(set_to_none=False), or accumulation over several backwards):
                # nothing consumes the lists meanwhile.  Bounded: beyond a few pending lists (each an int32 tensor of up
                # to min(points in voxel lists, cap * K) entries) the whole buffer is zeroed when it is next handed out
                if len(pending) >= self._MAX_PENDING_ROW_LISTS:
                    self._gdirty[key] = self._ALL_ROWS
                else:
                    pending.append((index, count))

    def _get_outputs_fused_grad(self, ray_bundle):
        """With autograd: pnr_render_views forwards (the reference's 0.3 jitter with a fresh seed per call; clamped only
        outside training, as nerfstudio's RGBRenderer), pnr_render_backward behind a torch.autograd.Function for d loss /
        d {points_embeding, points_color, points_dir, MLP weights} -- what autograd derives for studio_model.py:263-399.
        In training mode `conf_coefficient` (studio_model.py:288-292) is gathered here with torch ops so that its loss
        term reaches points_conf: the reference's tensor is [1,R'',SR,K] with unfilled slots reading point 0
        (studio_utils.py:193-199, clamp(pidx, 0)); the same MULTISET of values is returned in a fixed-size form --
        values [cap*K + 1] with integer multiplicities `conf_coefficient_weights` (1 for a filled slot, 0 for a slot of
        the workspace that holds nothing, and the number of unfilled slots of kept rays for the single entry of point
        0) -- so that no count has to reach the host; get_loss_dict takes the weighted mean, which is all the loss (a
        mean) looks at."""
        pose = self._device_pose(ray_bundle)
        cams, ray_cam = (None, None) if pose is not None else self._bundle_cameras(ray_bundle)
        rnd = self._fused_renderer(clamp=not self.training, live=True)
        rnd.opts.jitter = float(self.neural_points.jitter)
        rnd.opts.seed = self._next_seed()
        rnd.opts.early_stop_eps = 0.0
        rnd.tape = bool(getattr(self.config, "hip_tape_from_render", True))
        npts = self.neural_points
        mlp = []
        for name in MLP_TENSOR_ORDER:
            mod = self.get_submodule(name)
            mlp += [mod.weight, mod.bias]
        dirs = ray_bundle.directions.to(self._device).reshape(-1, 3)
        R = dirs.shape[0]
        cap = self._worst_case_cap(rnd, R, backward=True)
        if cap is None:
            self.host_reads += 1
        rgb, ray_mask = _FusedRenderFn.apply(self, rnd, dirs, cams if pose is None else pose, ray_cam, cap,
                                             npts.points_embeding, npts.points_color, npts.points_dir, *mlp)
        out = {"coarse_raycolor": rgb, "ray_mask": ray_mask}
        if self.training and getattr(self.config, "hip_conf_loss_kernel", True):
            term, both = _ConfLossFn.apply(rnd, float(self.config.zero_epsilon), npts.points_conf)
            out["conf_coefficient_loss_term"] = term
            out["conf_coefficient_slots"] = both[1]      # elements of the reference's [1,R'',SR,K] tensor
        elif self.training:
            cnt = rnd._counters_dev                                    # int64 [PNR_NUM_COUNTERS] on the device
            pidx = rnd.taps(R)["smp_pidx"]                             # [cap, K] view of the render workspace
            rows = torch.arange(pidx.shape[0], device=pidx.device)[:, None] < cnt[2]       # selected samples
            filled = (pidx >= 0) & rows
            conf = npts.points_conf[0, :, 0]
            # (rows of the workspace beyond the selected samples hold whatever was there: mask BEFORE indexing.  The
            # masked slots carry weight 0; they read DISTINCT rows -- slot j reads point j mod N -- because autograd's
            # index backward adds into the row of every slot, zero or not: two million atomic adds on ONE row take 30 ms)
            spread = torch.arange(pidx.numel(), device=pidx.device, dtype=torch.int32).remainder_(conf.shape[0])
            index = torch.where(filled.reshape(-1), pidx.reshape(-1), spread).long()
            values = torch.cat([conf[index], conf[0:1]])
            n_slots = cnt[1] * (self.config.SR * self.config.K)        # rays kept x SR x K: the reference's tensor
            w = torch.cat([filled.reshape(-1).to(torch.float32),
                           (n_slots - filled.sum()).to(torch.float32).reshape(1)])
            out["conf_coefficient"] = values - (values - torch.clamp(values, min=0.0001, max=1)).detach()
            out["conf_coefficient_weights"] = w
        return out

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
        single = getattr(self.config, "hip_single_camera_bundles", False)
        self.eval()
        try:
            self.config.hip_single_camera_bundles = False     # (the probe takes the cameras from the host)
            out = self._get_outputs_fused(ray_bundle)
            out.update(self._renderer.probe())
        finally:
            self.config.hip_single_camera_bundles = single
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
        c = self.config
        if self._fusable() and not torch.is_grad_enabled():
            return self._get_outputs_fused(ray_bundle)
        if self._fusable() and (getattr(c, "hip_fused_training", True) or not self.training):
            return self._get_outputs_fused_grad(ray_bundle)
        if self.unfused_outputs_fn is None:
            why = ("hip_fused_training is off" if self._fusable() else
                   "the fused kernels cover the default network shape only (32 features, 3/5/4 frequencies, "
                   "agg_dist_pers 20, 256/128 hidden units, 2+2+3 layers, colour and dir inputs, unit axis weights)")
            raise RuntimeError(f"PointNerf.get_outputs: {why}; this package holds no PyTorch-op render to fall back to "
                               f"(PointNerf.unfused_outputs_fn is the hook the comparison tests install)")
        return self.unfused_outputs_fn(self, ray_bundle)

    # ---- the rest of the plugin surface ---------------------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        """studio_model.py:401-413: `neural_points` = parameters named neural_points.points*, `fields` = the rest."""
        if self.mlp_base is None:
            raise ValueError("populate_fields() must be called before get_param_groups")
        named = list(self.named_parameters())
        return {"neural_points": [p for n, p in named if n.startswith("neural_points.points")],
                "fields": [p for n, p in named if not n.startswith("neural_points.points")]}

    def get_training_callbacks(self, training_callback_attributes: TrainingCallbackAttributes) -> List[TrainingCallback]:
        """The reference inherits nerfstudio's empty list.  Here one callback runs after every optimiser step: the MLP
        weights changed, so their PACKED copy (2.9 MB, MFMA operand order) is re-packed by the next render; the packed
        point rows are only marked stale for the next EVAL render -- training renders refresh the rows they read from
        the bound parameters (NeuralPoints.fused_scene(live=True)), so a step is followed by no O(N) work.  The voxel
        structure, the SceneHIP / RendererHIP objects and their workspaces stay: points_xyz is frozen
        (studio_utils.py:84), and a real change of it is caught by the (data_ptr, _version) key of fused_scene."""
        def _invalidate(step: int = 0):
            self.neural_points.invalidate_packed()
            self._weights_key = None
        return [TrainingCallback(where_to_run=[TrainingCallbackLocation.AFTER_TRAIN_ITERATION], func=_invalidate)]

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """studio_model.py:415-431."""
        device = outputs["coarse_raycolor"].device
        image = batch["image"].to(device)
        # the reference compacts both tensors with masked_select (a device-to-host read each: the sizes) and takes
        # MSELoss over what is left; the same mean as a masked sum, nothing read back
        loss_dict = {"ray_masked_coarse_raycolor_loss": _MaskedMSEFn.apply(outputs["coarse_raycolor"], image,
                                                                           outputs["ray_mask"])}
        if self.training and "conf_coefficient_loss_term" in outputs:
            # (the fused path computed the mean itself: _ConfLossFn)
            loss_dict["conf_coefficient_loss"] = outputs["conf_coefficient_loss_term"] * self.config.zero_one_loss_weights
        elif self.training:
            val = torch.clamp(outputs["conf_coefficient"], self.config.zero_epsilon, 1 - self.config.zero_epsilon)
            term = torch.log(val) + torch.log(1 - val)
            w = outputs.get("conf_coefficient_weights")
            # (the fused path returns the reference's values with multiplicities instead of repeating them: see
            # _get_outputs_fused_grad; the weighted mean IS the reference's torch.mean over its [1,R'',SR,K] tensor)
            mean = torch.mean(term) if w is None else torch.sum(term * w) / torch.sum(w)
            loss_dict["conf_coefficient_loss"] = mean * self.config.zero_one_loss_weights
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
