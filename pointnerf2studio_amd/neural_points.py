"""Host-side mirror of the reference's L2 components (pointnerf/nerfstudio/studio_utils.py), same names,
argument meaning and return layouts, backed by the HIP library:

  QueryWorldcoordsHIP.woord_query_grid_point_index  <- the pybind op of query_worldcoords.cpp:33-78
  PointNeRFEncoding                                 <- studio_utils.py:47-68
  NeuralPoints                                      <- studio_utils.py:71-209 (forward -> the same 13-tuple)

`NeuralPoints.forward` is the drop-in (compat) path: HIP query + PyTorch-ROCm gathers, differentiable w.r.t.
the point features exactly as in the reference.  `NeuralPoints.render` is the fused HIP path used by
`PointNerf.get_outputs` outside training.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .ns_compat import Encoding
from .renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters, query_raypos


class QueryWorldcoordsHIP:
    """Stands where the reference's JIT-built `query_worldcoords_cuda` module stands (studio_utils.py:77-82).

    The voxel structure is cached per (xyz storage, version, grid parameters): the reference rebuilds it on
    every call (query_worldcoords.cu:314-365); here a rebuild happens only when the cloud or the grid changes."""

    def __init__(self, compat_drop_voxel0: bool = True):
        self.compat_drop_voxel0 = compat_drop_voxel0
        self._scene: Optional[SceneHIP] = None
        self._key = None
        self.last_counters: Dict[str, int] = {}

    def scene_for(self, point_xyz_w_tensor, kernel_size, query_size, scaled_vdim, max_o, P, ranges, scaled_vsize):
        xyz = point_xyz_w_tensor
        tolist = lambda t: tuple(np.asarray(t.detach().cpu() if torch.is_tensor(t) else t).reshape(-1).tolist())
        key = (xyz.data_ptr(), xyz._version, tuple(xyz.shape), tolist(kernel_size), tolist(query_size),
               tolist(scaled_vdim), int(max_o), int(P), tolist(ranges), tolist(scaled_vsize))
        if self._scene is None or key != self._key:
            scene = SceneHIP()
            scene.build(xyz.reshape(-1, 3), tolist(ranges), tolist(scaled_vsize), tolist(scaled_vdim),
                        tolist(kernel_size), tolist(query_size), P, max_o, self.compat_drop_voxel0)
            self._scene, self._key = scene, key
        return self._scene

    def woord_query_grid_point_index(self, raypos_tensor, point_xyz_w_tensor, actual_numpoints_tensor, kernel_size,
                                     query_size, SR, K, R, D, scaled_vdim, max_o, P, radius_limit, ranges,
                                     scaled_vsize, kMaxThreadsPerBlock, NN):
        """Same 17 arguments as the reference op.  `actual_numpoints_tensor`, `kMaxThreadsPerBlock` and `NN`
        are accepted and unused (the reference ignores NN, cu:237; B is 1 in the plugin).  Returns
        [sample_pidx int32 [1,R'',SR,K], sample_loc f32 [1,R'',SR,3], ray_mask int8 [1,R]]."""
        if not raypos_tensor.is_cuda or not point_xyz_w_tensor.is_cuda:
            raise RuntimeError("woord_query_grid_point_index: tensors must be on the GPU (HIP-only op)")
        if point_xyz_w_tensor.dim() == 3 and point_xyz_w_tensor.shape[0] != 1:
            raise RuntimeError("woord_query_grid_point_index: batch size B must be 1")
        if tuple(raypos_tensor.shape[-3:-1]) != (R, D):
            raise RuntimeError(f"raypos has shape {tuple(raypos_tensor.shape)}, expected [1,{R},{D},3]")
        scene = self.scene_for(point_xyz_w_tensor, kernel_size, query_size, scaled_vdim, max_o, P, ranges,
                               scaled_vsize)
        radius = float(radius_limit.item() if torch.is_tensor(radius_limit) else radius_limit)
        pidx, loc, mask, counters = query_raypos(scene, raypos_tensor, int(SR), int(K), radius)
        self.last_counters = counters
        return [pidx, loc, mask]


class PointNeRFEncoding(Encoding):
    """studio_utils.py:47-68: x * 2^f, then interleaved (sin, cos) or, with ori, [x, sin(all), cos(all)]."""

    def __init__(self, in_dim: int, num_frequencies: int, ori: bool = False) -> None:
        super().__init__(in_dim)
        self.num_frequencies = num_frequencies
        self.ori = ori

    def forward(self, in_tensor, covs=None):
        freq_bands = (2 ** torch.arange(self.num_frequencies).float()).to(in_tensor.device)
        ori_c = in_tensor.shape[-1]
        pts = (in_tensor[..., None] * freq_bands).reshape(in_tensor.shape[:-1] + (self.num_frequencies * ori_c,))
        if self.ori:
            return torch.cat([in_tensor, torch.sin(pts), torch.cos(pts)], dim=-1)
        return torch.stack([torch.sin(pts), torch.cos(pts)], dim=-1).reshape(pts.shape[:-1] + (pts.shape[-1] * 2,))


def near_far_linear_ray_generation(campos, raydir, point_count, near=0.1, far=10, jitter=0., **kargs):
    """reference models/rendering/diff_ray_marching.py:292-336, device-agnostic restatement used by the compat
    path (the fused path never materialises raypos)."""
    dev = campos.device
    tvals = torch.linspace(0, 1, point_count + 1, device=dev).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]) * (
        1 + jitter * (torch.rand((raydir.shape[0], raydir.shape[1], point_count), device=dev) - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((end.shape[0], end.shape[1], 1), device=dev), end], dim=2)
    end = near + end
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    raypos = campos[:, None, None, :] + raydir[:, :, None, :] * mid[:, :, :, None]
    valid = torch.ones_like(mid)
    seg = seg * torch.linalg.norm(raydir[..., None, :], axis=-1)
    return raypos, seg, valid, mid


class NeuralPoints(nn.Module):
    """studio_utils.py:71-209.  Parameter names, shapes and requires_grad flags are the reference's, so
    `get_param_groups` (prefix `neural_points.points`) and checkpoints keep working."""

    def __init__(self, state_dict, device, config):
        super().__init__()
        self.config = config
        self.device = device
        self.query_worldcoords_cuda = QueryWorldcoordsHIP()   # attribute name kept for drop-in parity
        self.points_xyz = nn.Parameter(state_dict["neural_points.xyz"].to(device))
        self.points_embeding = nn.Parameter(state_dict["neural_points.points_embeding"].to(device))
        self.points_conf = nn.Parameter(state_dict["neural_points.points_conf"].to(device))
        self.points_dir = nn.Parameter(state_dict["neural_points.points_dir"].to(device))
        self.points_color = nn.Parameter(state_dict["neural_points.points_color"].to(device))
        self.points_Rw2c = nn.Parameter(state_dict["neural_points.Rw2c"].to(device))
        self.points_xyz.requires_grad = False
        self.points_embeding.requires_grad = config.feat_grad
        self.points_conf.requires_grad = config.conf_grad
        self.points_dir.requires_grad = config.dir_grad
        self.points_color.requires_grad = config.color_grad
        self.points_Rw2c.requires_grad = False

        self.reg_weight = 0.
        self.kernel_size = np.asarray(config.kernel_size, dtype=np.int32)
        self.kernel_size_tensor = torch.as_tensor(self.kernel_size, device=device, dtype=torch.int32)
        self.query_size = np.asarray(config.query_size, dtype=np.int32)
        self.query_size_tensor = torch.as_tensor(self.query_size, device=device, dtype=torch.int32)
        self.radius_limit_np = np.asarray(4 * max(config.vsize[0], config.vsize[1])).astype(np.float32)
        self.vscale_np = np.array(config.vscale, dtype=np.int32)
        self.scaled_vsize_np = (list(config.vsize) * self.vscale_np).astype(np.float32)
        self.scaled_vsize_tensor = torch.as_tensor(self.scaled_vsize_np, device=device)
        self.jitter = 0.3   # the reference hard-codes 0.3, train and eval (studio_utils.py:166)
        # fused path state
        self._fused_scene: Optional[SceneHIP] = None
        self._fused_key = None
        self._packed_key = None      # versions of the feature tensors at the last FULL pack (None: no rows yet)
        self._packed_stale = False   # an eval render must re-pack in full whatever the versions say
        self._bound_key = None

    # ---- reference helpers ---------------------------------------------------------------------------
    def get_hyperparameters(self, vsize_np, point_xyz_w_tensor, ranges=None):
        """studio_utils.py:115-127 (min/max on the device, the rest on the host with the reference's numpy
        promotions).  Cached per cloud version: the reference recomputes it on every forward."""
        xyz = point_xyz_w_tensor
        key = (xyz.data_ptr(), xyz._version, tuple(ranges))
        if getattr(self, "_hyp_key", None) != key:
            h = grid_hyperparameters(xyz.reshape(-1, 3), vsize_np, self.vscale_np, self.config.kernel_size, ranges)
            self._hyp = (torch.as_tensor(h.ranges, device=xyz.device), vsize_np, h.scaled_vdim, h)
            self._hyp_key = key
        return self._hyp[0], self._hyp[1], self._hyp[2]

    def w2pers(self, point_xyz, camrotc2w, campos):
        shift = point_xyz[None, ...] - campos[:, None, :]
        xyz = torch.sum(camrotc2w[:, None, :, :] * shift[:, :, :, None], dim=-2)
        return torch.stack([xyz[:, :, 0] / xyz[:, :, 2], xyz[:, :, 1] / xyz[:, :, 2], xyz[:, :, 2]], dim=-1)

    def w2pers_loc(self, point_xyz_w, camrotc2w, campos):
        shift = point_xyz_w - campos[:, None, :]
        xyz_c = torch.sum(shift[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, ...], dim=-1)
        z = xyz_c[..., 2]
        return torch.stack([xyz_c[..., 0] / z, xyz_c[..., 1] / z, z], dim=-1)

    def _camera(self, ray_bundle):
        rot = ray_bundle.metadata["camrotc2w"]
        if rot.shape[0] != 3:
            rot = rot[0].view(3, 3)
        return rot.unsqueeze(0).to(self.device), ray_bundle.origins[0].unsqueeze(0).to(self.device)

    # ---- compat path: the reference's forward, HIP query inside ----------------------------------------
    def forward(self, ray_bundle):
        cam_rot_tensor, cam_pos_tensor = self._camera(ray_bundle)
        ray_dirs_tensor = ray_bundle.directions.unsqueeze(0).to(self.device)
        near_depth, far_depth = ray_bundle.nears[0].item(), ray_bundle.fars[0].item()
        point_xyz_w_tensor = self.points_xyz[None, ...].detach()
        actual_numpoints_tensor = torch.ones([1], device=self.device, dtype=torch.int32) * point_xyz_w_tensor.shape[1]
        ranges_tensor, vsize_np, scaled_vdim_np = self.get_hyperparameters(self.config.vsize, self.points_xyz,
                                                                           ranges=self.config.ranges)
        raypos_tensor, _, _, _ = near_far_linear_ray_generation(cam_pos_tensor, ray_dirs_tensor,
                                                                self.config.z_depth_dim, near=near_depth,
                                                                far=far_depth, jitter=self.jitter)
        D, R = raypos_tensor.shape[2], ray_dirs_tensor.shape[1]
        sample_pidx_tensor, sample_loc_w_tensor, ray_mask_tensor = \
            self.query_worldcoords_cuda.woord_query_grid_point_index(
                raypos_tensor, point_xyz_w_tensor, actual_numpoints_tensor, self.kernel_size_tensor,
                self.query_size_tensor, self.config.SR, self.config.K, R, D,
                torch.as_tensor(scaled_vdim_np, device=self.device), self.config.max_o, self.config.P,
                torch.as_tensor(self.radius_limit_np, device=self.device), ranges_tensor, self.scaled_vsize_tensor,
                self.config.gpu_maxthr, self.config.NN)

        sample_ray_dirs_tensor = torch.masked_select(ray_dirs_tensor, ray_mask_tensor[..., None] > 0).reshape(
            1, -1, 3)[..., None, :].expand(-1, -1, self.config.SR, -1).contiguous()
        sample_pnt_mask = sample_pidx_tensor >= 0
        B, R, SR, K = sample_pidx_tensor.shape
        flat = torch.clamp(sample_pidx_tensor, min=0).view(-1).long()
        sample_loc_tensor = self.w2pers_loc(sample_loc_w_tensor, cam_rot_tensor, cam_pos_tensor)
        # perspective coordinates of the GATHERED neighbours only (the reference projects all N points and
        # concatenates an [1,N,38] table per call, studio_utils.py:197-199; same values, M rows instead of N)
        sampled_xyz = torch.index_select(self.points_xyz, 0, flat)
        sampled_xyz_pers = self.w2pers(sampled_xyz, cam_rot_tensor, cam_pos_tensor).view(B, R, SR, K, 3)
        sampled_xyz = sampled_xyz.view(B, R, SR, K, 3)
        sampled_embedding = torch.index_select(self.points_embeding, 1, flat).view(B, R, SR, K, -1)
        sampled_color = torch.index_select(self.points_color, 1, flat).view(B, R, SR, K, -1)
        sampled_dir = torch.index_select(self.points_dir, 1, flat).view(B, R, SR, K, -1)
        sampled_conf = torch.index_select(self.points_conf, 1, flat).view(B, R, SR, K, -1)
        return (sampled_color, self.points_Rw2c, sampled_dir, sampled_embedding, sampled_xyz_pers, sampled_xyz,
                sampled_conf, sample_loc_tensor, sample_loc_w_tensor, sample_pnt_mask, sample_ray_dirs_tensor,
                vsize_np, ray_mask_tensor)

    # ---- fused path --------------------------------------------------------------------------------------
    def _scene_key(self):
        return (self.points_xyz.data_ptr(), self.points_xyz._version, self.config.P, self.config.max_o)

    def fused_scene(self, live: bool = False) -> SceneHIP:
        """Voxel structure + packed point rows, rebuilt / repacked only when the tensors changed.
        live=True (training steps): the parameter tensors are BOUND to the scene (pnr_points_bind) and every render
        re-packs the rows of its own neighbour points from them -- an optimiser step is then followed by no O(N) re-pack
        at all (pnr_points_pack over 6 M points moves 2.3 GB; a 4096-ray batch reads ~60 k rows).  live=False (eval):
        one full re-pack when the features changed since the last one, then none."""
        _, _, scaled_vdim_np = self.get_hyperparameters(self.config.vsize, self.points_xyz, ranges=self.config.ranges)
        h = self._hyp[3]
        key = self._scene_key()
        if self._fused_scene is None or key != self._fused_key:
            scene = SceneHIP()
            scene.build(self.points_xyz.detach(), h.ranges, h.scaled_vsize, h.scaled_vdim, self.config.kernel_size,
                        self.config.query_size, self.config.P, self.config.max_o, True)
            self._fused_scene, self._fused_key, self._packed_key, self._bound_key = scene, key, None, None
        tensors = (self.points_embeding, self.points_conf, self.points_dir, self.points_color)
        pkey = tuple((p.data_ptr(), p._version) for p in tensors)
        scene = self._fused_scene
        if live and self._packed_key is not None:      # rows of this cloud exist: bind, never re-pack in full
            bkey = tuple(p.data_ptr() for p in tensors)
            if bkey != getattr(self, "_bound_key", None) or not scene.bound:
                scene.bind_points(self.points_xyz.detach(), self.points_embeding.detach(), self.points_conf.detach(),
                                  self.points_dir.detach(), self.points_color.detach())
                self._bound_key = bkey
            return scene
        if not live and scene.bound:
            scene.unbind_points()
            self._bound_key = None
        if pkey != self._packed_key or self._packed_stale:
            scene.pack_points(self.points_xyz.detach(), self.points_embeding.detach(), self.points_conf.detach(),
                              self.points_dir.detach(), self.points_color.detach())
            self._packed_key, self._packed_stale = pkey, False
        if live:
            return self.fused_scene(live=True)
        return scene

    # ---- point growing / pruning (reference models/neural_points/neural_points.py:341-393) ---------------------
    def _replace_points(self, xyz, emb, conf, pdir, color, old_index: torch.Tensor) -> None:
        """New Parameters (the cloud's size changed, so the optimiser has to be re-created by the caller, as the
        reference's trainer does: run/train_studio.py:676-684,714-716) and the voxel structure updated in place
        (pnr_scene_update) instead of rebuilt from nothing."""
        cfg = self.config
        # the scene may only be UPDATED if it was built on the cloud old_index refers to: pnr_scene_update reuses the
        # cell codes of surviving points.  A cloud edited in place since the last render (load_state_dict, copy_) has
        # a different key: then the structure is rebuilt from nothing at the next render.
        scene_is_current = self._fused_scene is not None and self._fused_key == self._scene_key()
        if self._fused_scene is not None:
            self._fused_scene.unbind_points()
        self._bound_key = None
        self.points_xyz = nn.Parameter(xyz.contiguous(), requires_grad=False)
        self.points_embeding = nn.Parameter(emb.contiguous(), requires_grad=bool(cfg.feat_grad))
        self.points_conf = nn.Parameter(conf.contiguous(), requires_grad=bool(cfg.conf_grad))
        self.points_dir = nn.Parameter(pdir.contiguous(), requires_grad=bool(cfg.dir_grad))
        self.points_color = nn.Parameter(color.contiguous(), requires_grad=bool(cfg.color_grad))
        self._packed_key = None
        self.query_worldcoords_cuda._key = None      # the compat op's own cache follows the cloud by key
        if scene_is_current and self.points_xyz.is_cuda:
            self._hyp_key = None
            self.get_hyperparameters(cfg.vsize, self.points_xyz, ranges=cfg.ranges)
            h = self._hyp[3]
            self._fused_scene.update(self.points_xyz.detach(), old_index, h.ranges, h.scaled_vsize, h.scaled_vdim,
                                     cfg.kernel_size, cfg.query_size, cfg.P, cfg.max_o, True)
            self._fused_key = (self.points_xyz.data_ptr(), self.points_xyz._version, cfg.P, cfg.max_o)
        else:
            self._fused_key = None

    @torch.no_grad()
    def prune(self, thresh: float) -> int:
        """neural_points.py:341-364: keeps the points with points_conf >= thresh.  Returns how many were removed."""
        mask = self.points_conf[0, ..., 0] >= thresh
        old = torch.nonzero(mask).reshape(-1).to(torch.int32)
        n_before = int(mask.shape[0])
        self._replace_points(self.points_xyz[mask, :], self.points_embeding[:, mask, :], self.points_conf[:, mask, :],
                             self.points_dir[:, mask, :], self.points_color[:, mask, :], old)
        return n_before - int(old.numel())

    @torch.no_grad()
    def grow_points(self, add_xyz, add_embedding, add_color, add_dir, add_conf) -> int:
        """neural_points.py:367-393: appends points (add_xyz [A,3], add_embedding [A,32], add_color [A,3], add_dir [A,3],
        add_conf [A,1]) behind the existing ones.  Returns the new point count."""
        dev = self.points_xyz.device
        N, A = self.points_xyz.shape[0], add_xyz.shape[0]
        old = torch.cat([torch.arange(N, dtype=torch.int32, device=dev),
                         torch.full((A,), -1, dtype=torch.int32, device=dev)])
        f = lambda t: t.to(device=dev, dtype=torch.float32)
        self._replace_points(torch.cat([self.points_xyz, f(add_xyz)], dim=0),
                             torch.cat([self.points_embeding, f(add_embedding)[None, ...]], dim=1),
                             torch.cat([self.points_conf, f(add_conf)[None, ...]], dim=1),
                             torch.cat([self.points_dir, f(add_dir)[None, ...]], dim=1),
                             torch.cat([self.points_color, f(add_color)[None, ...]], dim=1), old)
        return N + A

    def invalidate(self) -> None:
        """Forces a rebuild of the voxel structure + repack at the next fused render (the cloud itself changed)."""
        self._fused_key = None
        self._packed_key = None

    def invalidate_packed(self) -> None:
        """Forces a full repack of the point rows at the next EVAL render (features changed, positions did not).  A bound
        scene (training) needs none: its renders refresh the rows they read."""
        self._packed_stale = True
