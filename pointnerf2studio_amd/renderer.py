"""Thin tensor-level wrapper over the C ABI (include/pnr.h): PyTorch-ROCm tensors in, tensors out.

`SceneHIP`   -- persistent voxel structure + packed point table (pnr_scene_*)
`WeightsHIP` -- MLP weights in MFMA operand order (pnr_weights_*)
`query_raypos` -- the drop-in op behind `woord_query_grid_point_index`
`RendererHIP.render` -- NeuralPoints.forward + PointNerf.get_outputs fused (pnr_render)

PyTorch is plumbing here: device memory, streams, and nothing else.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib

MLP_TENSOR_ORDER = [
    "mlp_base.layers.0", "mlp_base.layers.1", "mlp_head.layers.0", "mlp_head.layers.1",
    "field_output_density.net",
    "mlp_color.layers.0", "mlp_color.layers.1", "mlp_color.layers.2", "field_output_color.net",
]
MLP_SHAPES = [(256, 284), (256, 256), (256, 263), (256, 256), (1, 256), (128, 280), (128, 128), (128, 128), (3, 128)]


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def coarse_segments(D: int, near: float, far: float) -> torch.Tensor:
    """tvals[j+1] - tvals[j] of near_far_linear_ray_generation (diff_ray_marching.py:307-311), float32 [D]."""
    tvals = torch.linspace(0, 1, D + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    return (tvals[..., 1:] - tvals[..., :-1]).reshape(D).contiguous()


def coarse_t_table(D: int, near: float, far: float) -> torch.Tensor:
    """Ray parameters of the D coarse mid-points at jitter 0, evaluated with the very torch ops of
    near_far_linear_ray_generation (reference models/rendering/diff_ray_marching.py:307-323) on the host."""
    tvals = torch.linspace(0, 1, D + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]).view(1, 1, D)
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((1, 1, 1)), end], dim=2)
    end = near + end
    return ((end[:, :, :-1] + end[:, :, 1:]) / 2).reshape(D).contiguous()


@dataclass
class GridHyper:
    """Output of NeuralPoints.get_hyperparameters (reference studio_utils.py:115-127)."""
    ranges: np.ndarray        # float32 [6]
    scaled_vsize: np.ndarray  # float32 [3]
    scaled_vdim: np.ndarray   # int32 [3]


def grid_hyperparameters(xyz: torch.Tensor, vsize: Sequence[float], vscale: Sequence[int],
                         kernel_size: Sequence[int], ranges: Sequence[float]) -> GridHyper:
    """Host-side mirror of get_hyperparameters, including the reference's numpy dtype promotions
    (the padding and the grid dims are evaluated in float64 from float32 inputs)."""
    vscale_np = np.array(vscale, dtype=np.int32)
    scaled_vsize_np = (list(vsize) * vscale_np).astype(np.float32)
    pts = xyz.detach().reshape(-1, 3)
    min_xyz = torch.min(pts, dim=0)[0].float().cpu()
    max_xyz = torch.max(pts, dim=0)[0].float().cpu()
    rmin = torch.as_tensor(list(ranges[:3]), dtype=torch.float32)
    rmax = torch.as_tensor(list(ranges[3:]), dtype=torch.float32)
    min_xyz = torch.max(torch.stack([min_xyz, rmin], 0), 0)[0]
    max_xyz = torch.min(torch.stack([max_xyz, rmax], 0), 0)[0]
    pad = torch.as_tensor(scaled_vsize_np * list(kernel_size) / 2, dtype=torch.float32)
    min_xyz = min_xyz - pad
    max_xyz = max_xyz + pad
    rng = torch.cat([min_xyz, max_xyz], dim=-1).numpy().astype(np.float32)
    vdim = (max_xyz - min_xyz).numpy() / list(vsize)
    scaled_vdim = np.ceil(vdim / vscale_np).astype(np.int32)
    return GridHyper(rng, scaled_vsize_np, scaled_vdim)


@dataclass
class View:
    """One pinhole view as nerfstudio's Cameras holds it: camera_to_worlds = [camrotc2w | campos], intrinsics in pixels."""
    campos: Sequence[float]
    camrotc2w: Sequence[float]     # 3x3, row-major
    fx: float
    fy: float
    cx: float
    cy: float
    near: float = 2.0
    far: float = 6.0

    @staticmethod
    def from_angle(campos, camrotc2w, H: int, W: int, camera_angle_x: float, near: float = 2.0, far: float = 6.0) -> "View":
        """nerf-synthetic convention: focal = 0.5 W / tan(0.5 camera_angle_x), principal point at the frame centre."""
        import math
        f = 0.5 * W / math.tan(0.5 * camera_angle_x)
        return View(campos, camrotc2w, f, f, 0.5 * W, 0.5 * H, near, far)


def _views_c(views: Sequence[View]):
    arr = (_lib.ViewC * len(views))()
    for i, v in enumerate(views):
        arr[i].campos[:] = [float(x) for x in torch.as_tensor(v.campos).reshape(3).tolist()]
        arr[i].camrotc2w[:] = [float(x) for x in torch.as_tensor(v.camrotc2w).reshape(9).tolist()]
        arr[i].near_plane, arr[i].far_plane = float(v.near), float(v.far)
        arr[i].fx, arr[i].fy, arr[i].cx, arr[i].cy = float(v.fx), float(v.fy), float(v.cx), float(v.cy)
    return arr


def pinhole_ray(view: View, x: int, y: int) -> np.ndarray:
    """Host statement of the direction the kernels generate for pixel (x, y) (pnr_pinhole_ray): float32 [3]."""
    out = (C.c_float * 3)()
    _lib.load().pnr_pinhole_ray(_views_c([view]), int(x), int(y), C.byref(out))
    return np.array(list(out), dtype=np.float32)


def camera_rays(views: Sequence[View], H: int, W: int, device, pixels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pnr_camera_rays: the directions pnr_render_camera generates in its kernels, written out as [n_views * n_pixels, 3]
    (view-major; `pixels` int32 flat ids on the device, None = every pixel of the frame row-major)."""
    lib = _lib.load()
    px = None if pixels is None else pixels.to(device=device, dtype=torch.int32).contiguous()
    if px is not None and px.dim() != 1:
        raise ValueError("camera_rays takes ONE flat pixel list for all views (pnr_camera_rays has no per-view form); "
                         "call it per view for per-view lists")
    n = H * W if px is None else px.numel()
    out = torch.empty((len(views) * n, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.pnr_camera_rays(_views_c(views), len(views), H, W, _ptr(px), n, _ptr(out), _stream_ptr(device)),
                   "pnr_camera_rays")
    return out


class SceneHIP:
    """Owns a pnr_scene_t.  Built once per point-cloud version; the reference rebuilds the equivalent
    structure for every ray chunk (query_worldcoords.cu:314-365)."""

    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.pnr_scene_create(C.byref(h)), "pnr_scene_create")
        self.handle = h
        self.device = None
        self.N = 0
        self.params = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pnr_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def build(self, xyz: torch.Tensor, ranges, scaled_vsize, scaled_vdim, kernel_size, query_size, P: int,
              max_o: int, compat_drop_voxel0: bool = True) -> Dict[str, int]:
        if not xyz.is_cuda:
            raise RuntimeError("SceneHIP.build: xyz must be a GPU tensor (the HIP path has no CPU fallback)")
        self.device = xyz.device
        pts = _f32c(xyz.reshape(-1, 3), self.device)
        self.N = pts.shape[0]
        gp = _lib.GridParams()
        gp.ranges[:] = [float(v) for v in np.asarray(ranges, dtype=np.float32)]
        gp.vox[:] = [float(v) for v in np.asarray(scaled_vsize, dtype=np.float32)]
        gp.dims[:] = [int(v) for v in scaled_vdim]
        gp.kernel_size[:] = [int(v) for v in kernel_size]
        gp.query_size[:] = [int(v) for v in query_size]
        gp.P, gp.max_o, gp.compat_drop_voxel0 = int(P), int(max_o), int(bool(compat_drop_voxel0))
        self.params = gp
        self._bound = None       # (the library drops bound tensors on a build / update)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pnr_scene_build(self.handle, _ptr(pts), self.N, C.byref(gp),
                                                _stream_ptr(self.device)), "pnr_scene_build")
        return self.info()

    def update(self, xyz: torch.Tensor, old_index: torch.Tensor, ranges, scaled_vsize, scaled_vdim, kernel_size,
               query_size, P: int, max_o: int, compat_drop_voxel0: bool = True) -> Dict[str, int]:
        """pnr_scene_update: the cloud changed (prune / grow).  old_index [N] int32: the index point i of the new
        cloud had in the previous one, -1 for an added point.  Rebuilds inside the memory the scene holds; surviving
        points keep their cell code when the grid is unchanged.  Same content as a fresh build().  The packed rows
        are stale afterwards: call pack_points."""
        if not xyz.is_cuda:
            raise RuntimeError("SceneHIP.update: xyz must be a GPU tensor")
        pts = _f32c(xyz.reshape(-1, 3), self.device)
        oi = old_index.to(device=self.device, dtype=torch.int32).contiguous()
        if oi.numel() != pts.shape[0]:
            raise ValueError(f"old_index has {oi.numel()} entries for {pts.shape[0]} points")
        gp = _lib.GridParams()
        gp.ranges[:] = [float(v) for v in np.asarray(ranges, dtype=np.float32)]
        gp.vox[:] = [float(v) for v in np.asarray(scaled_vsize, dtype=np.float32)]
        gp.dims[:] = [int(v) for v in scaled_vdim]
        gp.kernel_size[:] = [int(v) for v in kernel_size]
        gp.query_size[:] = [int(v) for v in query_size]
        gp.P, gp.max_o, gp.compat_drop_voxel0 = int(P), int(max_o), int(bool(compat_drop_voxel0))
        self._bound = None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pnr_scene_update(self.handle, _ptr(pts), pts.shape[0], C.byref(gp), _ptr(oi),
                                                 _stream_ptr(self.device)), "pnr_scene_update")
        self.N, self.params = pts.shape[0], gp
        return self.info()

    def update_info(self) -> Dict[str, int]:
        arr = (C.c_int64 * 4)()
        _lib.check(self.lib.pnr_scene_update_info(self.handle, C.byref(arr)), "pnr_scene_update_info")
        return dict(zip(["builds", "updates", "cells_reused", "scratch_bytes"], [int(v) for v in arr]))

    def info(self) -> Dict[str, int]:
        arr = (C.c_int64 * 8)()
        _lib.check(self.lib.pnr_scene_info(self.handle, C.byref(arr)), "pnr_scene_info")
        names = ["occupied_voxels", "max_o_overflow", "points_in_lists", "bricks", "device_bytes", "N",
                 "points_inside", "dropped_voxel_code"]
        return dict(zip(names, [int(v) for v in arr]))

    def pack_points(self, xyz, embedding, conf, direction, color) -> None:
        """Layouts of the reference's parameters (studio_utils.py:84-90): xyz [N,3], embedding [1,N,32],
        conf [1,N,1], dir [1,N,3], color [1,N,3]; leading singleton dims are ignored."""
        dev = self.device if self.device is not None else xyz.device
        if not xyz.is_cuda:
            raise RuntimeError("SceneHIP.pack_points: tensors must live on the GPU")
        self.device = dev
        x = _f32c(xyz.reshape(-1, 3), dev)
        N = x.shape[0]
        e = _f32c(embedding.reshape(N, -1), dev)
        if e.shape[1] != 32:
            raise ValueError(f"point_features_dim must be 32, got {e.shape[1]}")
        c = None if conf is None else _f32c(conf.reshape(N), dev)
        d = _f32c(direction.reshape(N, 3), dev)
        col = _f32c(color.reshape(N, 3), dev)
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_points_pack(self.handle, _ptr(x), _ptr(e), _ptr(c), _ptr(d), _ptr(col), N,
                                                _stream_ptr(dev)), "pnr_points_pack")
        # no synchronisation: converted temporaries return to torch's caching allocator, which hands a block out
        # again only to work queued behind this launch on the same stream

    @staticmethod
    def _raw(xyz, embedding, conf, direction, color, dev):
        """The five tensors as the C ABI wants them (float32, contiguous, leading singleton dims dropped).  For
        Parameters in the reference's layouts these are VIEWS of the same storage, not copies."""
        x = _f32c(xyz.reshape(-1, 3), dev)
        N = x.shape[0]
        e = _f32c(embedding.reshape(N, -1), dev)
        if e.shape[1] != 32:
            raise ValueError(f"point_features_dim must be 32, got {e.shape[1]}")
        c = None if conf is None else _f32c(conf.reshape(N), dev)
        return x, e, c, _f32c(direction.reshape(N, 3), dev), _f32c(color.reshape(N, 3), dev), N

    def pack_point_rows(self, xyz, embedding, conf, direction, color, index: torch.Tensor,
                        count: Optional[torch.Tensor] = None) -> None:
        """pnr_points_pack_rows: re-packs only the rows `index` (int32, device); `count` (int64 device scalar) limits the
        list to its first min(count, len(index)) entries without a host read."""
        dev = self.device
        x, e, c, d, col, N = self._raw(xyz, embedding, conf, direction, color, dev)
        idx = index.to(device=dev, dtype=torch.int32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_points_pack_rows(self.handle, _ptr(x), _ptr(e), _ptr(c), _ptr(d), _ptr(col), N,
                                                     _ptr(idx), idx.numel(), _ptr(count), _stream_ptr(dev)),
                       "pnr_points_pack_rows")

    def bind_points(self, xyz, embedding, conf, direction, color) -> None:
        """pnr_points_bind: every following render re-packs the rows of its distinct neighbour points from these LIVE
        tensors (a training loop: no O(N) re-pack after an optimiser step).  The tensors must be float32 and
        contiguous -- views of them are bound, not copies -- and are kept alive by this object."""
        dev = self.device
        x, e, c, d, col, N = self._raw(xyz, embedding, conf, direction, color, dev)
        for t, src in ((x, xyz), (e, embedding), (d, direction), (col, color)) + (() if conf is None else ((c, conf),)):
            if t.data_ptr() != src.data_ptr():
                raise ValueError("bind_points needs float32 contiguous GPU tensors (a copy would go stale)")
        _lib.check(self.lib.pnr_points_bind(self.handle, _ptr(x), _ptr(e), _ptr(c), _ptr(d), _ptr(col), N),
                   "pnr_points_bind")
        self._bound = (x, e, c, d, col)

    def unbind_points(self) -> None:
        if getattr(self, "_bound", None) is not None:
            _lib.check(self.lib.pnr_points_bind(self.handle, None, None, None, None, None, 0), "pnr_points_bind")
            self._bound = None

    @property
    def bound(self) -> bool:
        return getattr(self, "_bound", None) is not None


class WeightsHIP:
    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.pnr_weights_create(C.byref(h)), "pnr_weights_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pnr_weights_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def update(self, state: Dict[str, torch.Tensor], device, precision: Optional[str] = None) -> None:
        """pnr_weights_update: the layers changed, Rw2c did not -- one launch, no synchronisation (a training loop after
        every optimiser step).  precision: re-pack only the forms that mode reads (None = both)."""
        ws, bs = [], []
        for name, shape in zip(MLP_TENSOR_ORDER, MLP_SHAPES):
            ws.append(_f32c(state[name + ".weight"], device))
            bs.append(_f32c(state[name + ".bias"], device))
            if tuple(ws[-1].shape) != shape:
                raise ValueError(f"{name}: expected weight {shape}, got {tuple(ws[-1].shape)}")
        wp = (C.c_void_p * 9)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * 9)(*[b.data_ptr() for b in bs])
        with torch.cuda.device(device):
            _lib.check(self.lib.pnr_weights_update(self.handle, C.byref(wp), C.byref(bp),
                                                   -1 if precision is None else _lib.PRECISION[precision],
                                                   _stream_ptr(device)), "pnr_weights_update")

    def pack(self, state: Dict[str, torch.Tensor], Rw2c: torch.Tensor, device) -> None:
        """state: '<module>.weight' / '<module>.bias' for the nine Linear layers of MLP_TENSOR_ORDER
        (nerfstudio MLP / FieldHead naming, studio_model.py:193-221)."""
        ws, bs = [], []
        for name, shape in zip(MLP_TENSOR_ORDER, MLP_SHAPES):
            w = _f32c(state[name + ".weight"], device)
            b = _f32c(state[name + ".bias"], device)
            if tuple(w.shape) != shape or tuple(b.shape) != (shape[0],):
                raise ValueError(f"{name}: expected weight {shape}, got {tuple(w.shape)} / bias {tuple(b.shape)}")
            ws.append(w)
            bs.append(b)
        r = _f32c(Rw2c.reshape(3, 3), device)
        wp = (C.c_void_p * 9)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * 9)(*[b.data_ptr() for b in bs])
        with torch.cuda.device(device):
            _lib.check(self.lib.pnr_weights_pack(self.handle, C.byref(wp), C.byref(bp), _ptr(r),
                                                 _stream_ptr(device)), "pnr_weights_pack")


def query_raypos(scene: SceneHIP, raypos: torch.Tensor, SR: int, K: int, radius_limit: float):
    """woord_query_grid_point_index (reference query_worldcoords.cpp:33-78) on a built scene.
    raypos [1,R,D,3] -> (sample_pidx int32 [1,R'',SR,K], sample_loc f32 [1,R'',SR,3], ray_mask int8 [1,R],
    counters dict).  One host sync to learn R'' -- the reference has three (cu:310,382,426)."""
    lib, dev = scene.lib, raypos.device
    rp = _f32c(raypos, dev)
    R, D = rp.shape[-3], rp.shape[-2]
    pidx = torch.empty((max(R, 1), SR, K), dtype=torch.int32, device=dev)
    loc = torch.empty((max(R, 1), SR, 3), dtype=torch.float32, device=dev)
    mask = torch.zeros((max(R, 1),), dtype=torch.int8, device=dev)
    counters = torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev)
    if R == 0:
        return pidx[None, :0], loc[None, :0], mask[None, :0], dict(zip(_lib.COUNTER_NAMES, [0] * 8))
    nbytes = lib.pnr_query_workspace_bytes(R, D, SR, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_query_raypos(scene.handle, _ptr(rp), R, D, SR, K, float(radius_limit), _ptr(pidx),
                                        _ptr(loc), _ptr(mask), _ptr(counters), _ptr(ws), nbytes, _stream_ptr(dev)),
                   "pnr_query_raypos")
    cnt = counters.cpu().tolist()
    n = cnt[1]
    return pidx[None, :n], loc[None, :n], mask[None, :R], dict(zip(_lib.COUNTER_NAMES, cnt))


class RendererHIP:
    """Fused render of one ray bundle.  Owns a growable workspace; `cap_samples` (selected shading samples
    the workspace can hold) grows automatically when a frame overflows it.  precision: "fp32" (default: every product
    and sum in fp32 on v_mfma_f32_32x32x2_f32, the reference's arithmetic) or the opt-in "bf16x3" (three bf16 MFMA
    products per fp32 product, image within ~2e-5 of fp32)."""

    def __init__(self, scene: SceneHIP, weights: WeightsHIP, SR: int = 80, K: int = 8, D: int = 400,
                 radius_limit: float = 0.016, vsize_z: float = 0.004, eval_clamp: bool = True,
                 bg=(1.0, 1.0, 1.0), precision: str = "fp32", jitter: float = 0.0, seed: int = 0,
                 early_stop_eps: float = 0.0, tape: bool = False):
        self.lib = _lib.load()
        self.scene, self.weights = scene, weights
        self.opts = _lib.RenderOpts()
        self.opts.SR, self.opts.K, self.opts.D = int(SR), int(K), int(D)
        self.opts.radius_limit = float(np.float32(radius_limit))
        self.opts.vsize_z = float(np.float32(vsize_z))
        self.opts.eval_clamp = int(bool(eval_clamp))
        self.opts.bg[:] = [float(b) for b in bg]
        if precision not in _lib.PRECISION:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISION)}, got {precision!r}")
        self.opts.precision = _lib.PRECISION[precision]
        self.opts.jitter = float(jitter)
        self.opts.seed = int(seed) & 0xFFFFFFFF
        # 0: every sample with a neighbour is shaded (the reference's sample set); > 0: early ray termination
        self.opts.early_stop_eps = float(early_stop_eps)
        # tape: renders leave the activations of the four per-pair layers in the training workspace as they compute them
        # (pnr_render_opts_t.d_tape) and backward() does not recompute them.  For renders that ARE followed by a backward:
        # a taped render is a few per cent slower and allocates the training workspace.
        self.tape = bool(tape)
        self._tws = None
        self._ws = None
        self._ws_key = None
        self._tmid = {}
        self._cam_cache = {}
        self.cap_samples = 0
        self.calls = 0          # render calls so far (a backward must follow ITS render directly)
        self._last = None
        self.last_counters = None
        self._counters_dev = None

    def _workspace(self, R: int, cap: int, dev):
        key = (R, cap, self.opts.K)
        nbytes = self.lib.pnr_render_workspace_bytes_for(self.scene.handle, C.byref(self.opts), R, cap)
        if nbytes == 0:
            raise RuntimeError("pnr_render_workspace_bytes_for: " + self.lib.pnr_last_error().decode())
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # (a larger buffer serves a smaller call: the carving depends on (R, cap, K) only)
        self._ws_key = key
        self.cap_samples = cap
        if self.tape:
            tws = self._train_workspace(cap, dev)
            self.opts.d_tape, self.opts.tape_bytes = tws.data_ptr(), tws.numel()
        else:
            self.opts.d_tape, self.opts.tape_bytes = None, 0
        return self._ws

    def _train_workspace(self, cap: int, dev) -> torch.Tensor:
        nbytes = self.lib.pnr_backward_workspace_bytes(cap, self.opts.K)
        if self._tws is None or self._tws.numel() != nbytes:
            # (exactly the size of this capacity: the backward recognises the tape a render filled by pointer AND size)
            self._tws = None
            self._tws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._tws

    def tmid(self, near: float, far: float, dev) -> torch.Tensor:
        key = (self.opts.D, float(near), float(far), str(dev))
        if key not in self._tmid:
            self._tmid[key] = torch.stack([coarse_t_table(self.opts.D, float(near), float(far)),
                                           coarse_segments(self.opts.D, float(near), float(far))]).to(dev)
        return self._tmid[key]

    def render(self, directions: torch.Tensor, campos, camrotc2w, near: float, far: float,
               cap_samples: Optional[int] = None, sync_counters: bool = True, out: Optional[dict] = None):
        """One ray bundle of ONE camera (the reference's contract).  directions [R,3] (GPU), campos [3],
        camrotc2w [3,3] (host or device).  Returns dict with rgb [R,3], depth [R], acc [R], ray_mask [R] int8,
        counters (dict if sync_counters else only `counters_dev`)."""
        return self.render_views(directions, [(campos, camrotc2w, near, far)], directions.reshape(-1, 3).shape[0],
                                 cap_samples=cap_samples, sync_counters=sync_counters, out=out)

    def render_views(self, directions: torch.Tensor, cams, rays_per_cam: int, ray_cam: Optional[torch.Tensor] = None,
                     cap_samples: Optional[int] = None, sync_counters: bool = True, out: Optional[dict] = None):
        """Several cameras in one call (pnr_render_views).  `cams` = [(campos, camrotc2w, near, far), ...]; ray r
        belongs to camera ray_cam[r] (int32 GPU tensor) or, without it, to camera r // rays_per_cam."""
        dev = directions.device
        if not directions.is_cuda:
            raise RuntimeError("RendererHIP.render: directions must be a GPU tensor (no CPU fallback)")
        d = _f32c(directions.reshape(-1, 3), dev)
        R = d.shape[0]
        n = len(cams)
        if not 1 <= n <= _lib.MAX_CAMS:
            raise ValueError(f"1..{_lib.MAX_CAMS} cameras per call, got {n}")
        key = tuple((tuple(torch.as_tensor(c[0]).reshape(3).tolist()), tuple(torch.as_tensor(c[1]).reshape(9).tolist()),
                     float(c[2]), float(c[3])) for c in cams)
        cached = self._cam_cache.get(key)
        if cached is None:
            arr = (_lib.CameraC * n)()
            for i, (pos, rot, near, far) in enumerate(key):
                arr[i].campos[:] = pos
                arr[i].camrotc2w[:] = rot
                arr[i].near_plane, arr[i].far_plane = near, far
            tm = torch.stack([self.tmid(c[2], c[3], dev) for c in key]).contiguous()
            if len(self._cam_cache) > 64:
                self._cam_cache.clear()
            cached = self._cam_cache[key] = (arr, tm)
        arr, tm = cached
        rc = None if ray_cam is None else ray_cam.to(device=dev, dtype=torch.int32).contiguous()
        cap = int(cap_samples or self.cap_samples or max(4096, min(R * self.opts.SR, R * 16)))
        if out is None:
            out = {
                "rgb": torch.empty((R, 3), dtype=torch.float32, device=dev),
                "depth": torch.empty((R,), dtype=torch.float32, device=dev),
                "acc": torch.empty((R,), dtype=torch.float32, device=dev),
                "ray_mask": torch.empty((R,), dtype=torch.int8, device=dev),
                "counters_dev": torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev),
            }
        while True:
            ws = self._workspace(R, cap, dev)
            self._last = (d, R, arr, n, rc, int(rays_per_cam), cap)
            self.calls += 1
            with torch.cuda.device(dev):
                _lib.check(self.lib.pnr_render_views(
                    self.scene.handle, self.weights.handle, _ptr(d), R, arr, n, _ptr(rc), int(rays_per_cam), _ptr(tm),
                    C.byref(self.opts), _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out["acc"]), _ptr(out["ray_mask"]),
                    _ptr(out["counters_dev"]), _ptr(ws), ws.numel(), cap, _stream_ptr(dev)), "pnr_render_views")
            self._counters_dev = out["counters_dev"]
            if not sync_counters:
                self.last_counters = None      # (stale counters of an earlier call must not size anything)
                return out
            cnt = out["counters_dev"].cpu().tolist()
            out["counters"] = dict(zip(_lib.COUNTER_NAMES, cnt))
            self.last_counters = out["counters"]
            if cnt[6] == 0:
                return out
            # overflow: grow to what the frame actually needs (+12 %) and render again
            cap = int(cnt[2] * 1.125) + 1024

    def render_pose(self, directions: torch.Tensor, campos_dev: torch.Tensor, camrot_dev: torch.Tensor, near: float,
                    far: float, cap_samples: Optional[int] = None, sync_counters: bool = True,
                    out: Optional[dict] = None):
        """pnr_render_pose: one camera whose pose lives on the DEVICE (campos_dev [3], camrot_dev [9], float32 contiguous
        GPU tensors -- a bundle's origins[0] and camrotc2w): nothing is read back to launch the render.  backward() works
        after it (the camera stays in the workspace); probe() does not."""
        dev = directions.device
        if not directions.is_cuda:
            raise RuntimeError("RendererHIP.render_pose: directions must be a GPU tensor (no CPU fallback)")
        d = _f32c(directions.reshape(-1, 3), dev)
        pos, rot = _f32c(campos_dev.reshape(3), dev), _f32c(camrot_dev.reshape(9), dev)
        R = d.shape[0]
        tm = self.tmid(near, far, dev).reshape(1, 2, -1).contiguous()
        cap = int(cap_samples or self.cap_samples or max(4096, min(R * self.opts.SR, R * 16)))
        if out is None:
            out = {
                "rgb": torch.empty((R, 3), dtype=torch.float32, device=dev),
                "depth": torch.empty((R,), dtype=torch.float32, device=dev),
                "acc": torch.empty((R,), dtype=torch.float32, device=dev),
                "ray_mask": torch.empty((R,), dtype=torch.int8, device=dev),
                "counters_dev": torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev),
            }
        while True:
            ws = self._workspace(R, cap, dev)
            self._last = (d, R, None, 1, None, R, cap)      # (no host camera: the backward reads the workspace's)
            self._pose_keep = (pos, rot)
            self.calls += 1
            with torch.cuda.device(dev):
                _lib.check(self.lib.pnr_render_pose(
                    self.scene.handle, self.weights.handle, _ptr(d), R, _ptr(pos), _ptr(rot), float(near), float(far),
                    _ptr(tm), C.byref(self.opts), _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out["acc"]),
                    _ptr(out["ray_mask"]), _ptr(out["counters_dev"]), _ptr(ws), ws.numel(), cap, _stream_ptr(dev)),
                    "pnr_render_pose")
            self._counters_dev = out["counters_dev"]
            if not sync_counters:
                self.last_counters = None
                return out
            cnt = out["counters_dev"].cpu().tolist()
            out["counters"] = dict(zip(_lib.COUNTER_NAMES, cnt))
            self.last_counters = out["counters"]
            if cnt[6] == 0:
                return out
            cap = int(cnt[2] * 1.125) + 1024

    def render_camera(self, views: Sequence[View], H: int, W: int, pixels: Optional[torch.Tensor] = None,
                      cap_samples: Optional[int] = None, sync_counters: bool = True, out: Optional[dict] = None):
        """pnr_render_camera: the views' rays are generated inside the kernels from pose + intrinsics; no direction
        tensor is handed over.  `pixels` (int32 flat ids y * W + x on the device: what a tile shard owns) or None for
        the whole frame; the outputs have len(views) * n_pixels rows, view-major in `pixels` order.  A 2-D `pixels`
        [len(views), n_pixels] gives every view its own list (pnr_render_camera_lists)."""
        dev = self.scene.device
        n = len(views)
        if not 1 <= n <= _lib.MAX_CAMS:
            raise ValueError(f"1..{_lib.MAX_CAMS} views per call, got {n}")
        px = None if pixels is None else pixels.to(device=dev, dtype=torch.int32).contiguous()
        per_view = px is not None and px.dim() == 2
        if per_view and px.shape[0] != n:
            raise ValueError(f"per-view pixel lists: {px.shape[0]} lists for {n} views")
        n_px = H * W if px is None else int(px.shape[-1])
        R = n * n_px
        fn, who = ((self.lib.pnr_render_camera_lists, "pnr_render_camera_lists") if per_view
                   else (self.lib.pnr_render_camera, "pnr_render_camera"))
        key = ("views",) + tuple(id(v) for v in views)
        cached = self._cam_cache.get(key)
        # the whole value of every view: a View mutated in place (pose included) is packed again
        sig = [(v.fx, v.fy, v.cx, v.cy, v.near, v.far, tuple(torch.as_tensor(v.campos).reshape(3).tolist()),
                tuple(torch.as_tensor(v.camrotc2w).reshape(9).tolist())) for v in views]
        if cached is None or cached[3] != sig:
            arr = _views_c(views)
            tm = torch.stack([self.tmid(v.near, v.far, dev) for v in views]).contiguous()
            cams = (_lib.CameraC * n)()     # the same cameras without intrinsics: what a backward call takes
            for i in range(n):
                cams[i].campos[:] = list(arr[i].campos)
                cams[i].camrotc2w[:] = list(arr[i].camrotc2w)
                cams[i].near_plane, cams[i].far_plane = arr[i].near_plane, arr[i].far_plane
            if len(self._cam_cache) > 64:
                self._cam_cache.clear()
            # (the View objects are kept alive by the entry, so their ids stay theirs)
            cached = self._cam_cache[key] = (arr, tm, cams, sig, list(views))
        arr, tm, cams = cached[0], cached[1], cached[2]
        cap = int(cap_samples or self.cap_samples or max(4096, min(R * self.opts.SR, R * 16)))
        if out is None:
            out = {
                "rgb": torch.empty((R, 3), dtype=torch.float32, device=dev),
                "depth": torch.empty((R,), dtype=torch.float32, device=dev),
                "acc": torch.empty((R,), dtype=torch.float32, device=dev),
                "ray_mask": torch.empty((R,), dtype=torch.int8, device=dev),
                "counters_dev": torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev),
            }
        while True:
            ws = self._workspace(R, cap, dev)
            # a backward after this render reads the directions k_expand wrote into the workspace (taps.ray_dirs)
            self._last = ("camera", R, cams, n, None, n_px, cap)
            self.calls += 1
            with torch.cuda.device(dev):
                _lib.check(fn(
                    self.scene.handle, self.weights.handle, arr, n, int(H), int(W), _ptr(px), n_px, _ptr(tm),
                    C.byref(self.opts), _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out["acc"]), _ptr(out["ray_mask"]),
                    _ptr(out["counters_dev"]), _ptr(ws), ws.numel(), cap, _stream_ptr(dev)), who)
            self._counters_dev = out["counters_dev"]
            if not sync_counters:
                self.last_counters = None
                return out
            cnt = out["counters_dev"].cpu().tolist()
            out["counters"] = dict(zip(_lib.COUNTER_NAMES, cnt))
            self.last_counters = out["counters"]
            if cnt[6] == 0:
                return out
            cap = int(cnt[2] * 1.125) + 1024

    def backward(self, grad_rgb: torch.Tensor, state: Dict[str, torch.Tensor], num_points: int,
                 point_grads: bool = True, weight_grads: bool = True, sparse_points: bool = False,
                 into: Optional[Dict[str, Optional[torch.Tensor]]] = None) -> Dict[str, torch.Tensor]:
        """Gradients of the LAST render / render_views call (pnr_render_backward): d loss / d {embedding [N,32],
        color [N,3], dir [N,3], '<module>.weight', '<module>.bias'} for grad_rgb = d loss / d rgb [R,3].  `state` holds
        the raw MLP tensors the weights were packed from.  What torch autograd derives for studio_model.py:263-399;
        the MLP forward is recomputed in the renderer's precision ('rgb' in the result is that recomputed image; fp32:
        gradients agree with fp32 autograd to ~1e-6, bf16x3: the GEMMs on bf16 hi/lo splits).
        sparse_points: instead of the three dense [N, .] tensors, 'point_index' [U] (int64, ascending) and 'point_grads'
        [U, 40] = [d embedding | d color | d dir | 0 0] for the U distinct neighbour points of the render.
        into: {'embedding': [N*32], 'color': [N*3], 'dir': [N*3]} float32 contiguous tensors (any of them None) the point
        gradients are ACCUMULATED into (+= on the rows of the touched points only, nothing is allocated or zero-filled:
        what a training loop hands over, its .grad tensors or persistent buffers).
        The point gradients are summed in a fixed order: repeated calls return the same bits."""
        if getattr(self, "_last", None) is None:
            raise RuntimeError("RendererHIP.backward: no render call to differentiate")
        d, R, arr, n, rc, rays_per_cam, cap = self._last
        if isinstance(d, str):   # after render_camera: the per-hit-ray directions live in the render workspace
            d = self.taps(R)["ray_dirs"]
        dev = d.device
        g = _f32c(grad_rgb.reshape(R, 3), dev)
        ws_t, bs_t = [], []
        for name, shape in zip(MLP_TENSOR_ORDER, MLP_SHAPES):
            ws_t.append(_f32c(state[name + ".weight"], dev))
            bs_t.append(_f32c(state[name + ".bias"], dev))
        wp = (C.c_void_p * 9)(*[w.data_ptr() for w in ws_t])
        bp = (C.c_void_p * 9)(*[b.data_ptr() for b in bs_t])
        out: Dict[str, torch.Tensor] = {"rgb": torch.empty((R, 3), dtype=torch.float32, device=dev)}
        grads = _lib.GradsC()
        if point_grads and into is not None:
            for key, width, field in (("embedding", 32, "d_embedding"), ("color", 3, "d_color"), ("dir", 3, "d_dir")):
                t = into.get(key)
                if t is None:
                    continue
                if (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != num_points * width
                        or t.device != dev):
                    raise ValueError(f"backward(into=...): '{key}' must be a contiguous float32 tensor of "
                                     f"{num_points * width} elements on {dev}")
                setattr(grads, field, t.data_ptr())
        elif point_grads and sparse_points:
            # U of THIS render: from its synced counters, else read from the device (one host read; a stale count of
            # an earlier call would silently drop rows)
            U = (int(self.last_counters["points_unique"]) if self.last_counters is not None
                 else int(self._counters_dev[7].item()))
            out["point_grads"] = torch.empty((max(U, 1), 40), dtype=torch.float32, device=dev)
            sp_index = torch.empty((max(U, 1),), dtype=torch.int32, device=dev)
            grads.d_point_grads, grads.d_point_index, grads.point_cap = (out["point_grads"].data_ptr(),
                                                                         sp_index.data_ptr(), max(U, 1))
        elif point_grads:
            out["embedding"] = torch.zeros((num_points, 32), dtype=torch.float32, device=dev)
            out["color"] = torch.zeros((num_points, 3), dtype=torch.float32, device=dev)
            out["dir"] = torch.zeros((num_points, 3), dtype=torch.float32, device=dev)
            grads.d_embedding, grads.d_color, grads.d_dir = (out["embedding"].data_ptr(), out["color"].data_ptr(),
                                                             out["dir"].data_ptr())
        if weight_grads:
            # one zero-filled buffer, viewed per tensor (64-float aligned slices)
            sizes = [((sh[0] * sh[1] + 63) // 64 * 64, (sh[0] + 63) // 64 * 64) for sh in MLP_SHAPES]
            flat = torch.zeros(sum(a + b for a, b in sizes), dtype=torch.float32, device=dev)
            off = 0
            for i, (name, shape) in enumerate(zip(MLP_TENSOR_ORDER, MLP_SHAPES)):
                out[name + ".weight"] = flat[off:off + shape[0] * shape[1]].view(shape)
                off += sizes[i][0]
                out[name + ".bias"] = flat[off:off + shape[0]]
                off += sizes[i][1]
                grads.d_w[i] = out[name + ".weight"].data_ptr()
                grads.d_b[i] = out[name + ".bias"].data_ptr()
        self._train_workspace(cap, dev)
        ws = self._ws
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_render_backward(
                self.scene.handle, self.weights.handle, C.byref(wp), C.byref(bp), _ptr(d), R, arr, n, _ptr(rc),
                rays_per_cam, C.byref(self.opts), _ptr(g), _ptr(ws), ws.numel(), cap, _ptr(self._tws),
                self._tws.numel(), C.byref(grads), _ptr(out["rgb"]), _stream_ptr(dev)), "pnr_render_backward")
        # no synchronisation (see pack_points): everything is queued on torch's current stream
        if point_grads and sparse_points and into is None:
            out["point_grads"] = out["point_grads"][:U]
            out["point_index"] = sp_index[:U].to(torch.long)
        return out

    def conf_loss(self, conf: torch.Tensor, eps: float) -> torch.Tensor:
        """pnr_conf_loss on the LAST render: [mean(log v + log(1 - v)), element count] of the reference's conf_coefficient
        tensor (studio_model.py:288-292,427-429) as a float32 [2] device tensor; no host read."""
        if self._last is None:
            raise RuntimeError("RendererHIP.conf_loss: no render call yet")
        R, cap = self._last[1], self._last[6]
        dev = self.scene.device
        c = _f32c(conf.reshape(-1), dev)
        if getattr(self, "_conf_scratch", None) is None:
            self._conf_scratch = torch.empty(self.lib.pnr_conf_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ws = self._ws
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_conf_loss(self.scene.handle, C.byref(self.opts), R, _ptr(ws), ws.numel(), cap, _ptr(c),
                                              float(eps), _ptr(self._conf_scratch), _ptr(out), _stream_ptr(dev)),
                       "pnr_conf_loss")
        return out

    def conf_loss_backward(self, conf: torch.Tensor, eps: float, fwd: torch.Tensor, upstream: torch.Tensor,
                           grad_conf: torch.Tensor) -> None:
        """pnr_conf_loss_backward: grad_conf [N] (float32, contiguous) += d mean / d conf * upstream (device scalar)."""
        R, cap = self._last[1], self._last[6]
        dev = self.scene.device
        c = _f32c(conf.reshape(-1), dev)
        up = _f32c(upstream.reshape(1), dev)
        ws = self._ws
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_conf_loss_backward(self.scene.handle, C.byref(self.opts), R, _ptr(ws), ws.numel(), cap,
                                                       _ptr(c), float(eps), _ptr(self._conf_scratch), _ptr(fwd), _ptr(up),
                                                       _ptr(grad_conf), _stream_ptr(dev)), "pnr_conf_loss_backward")

    def touched(self, index: Optional[torch.Tensor] = None, count: Optional[torch.Tensor] = None):
        """pnr_render_touched: the distinct neighbour points of the LAST render (ascending), without a host read:
        (index int32 [cap] -- entries beyond the count repeat the first one --, count int64 [1] on the device).  cap =
        min(points in voxel lists, cap_samples * K) unless a buffer is passed in."""
        if self._last is None:
            raise RuntimeError("RendererHIP.touched: no render call yet")
        R, cap = self._last[1], self._last[6]
        dev = self.scene.device
        if index is None:
            u_cap = max(1, min(int(self.scene.info()["points_in_lists"]), cap * self.opts.K))
            index = torch.empty((u_cap,), dtype=torch.int32, device=dev)
        if count is None:
            count = torch.empty((1,), dtype=torch.int64, device=dev)
        ws = self._ws
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_render_touched(self.scene.handle, C.byref(self.opts), R, _ptr(ws), ws.numel(), cap,
                                                   _ptr(index), index.numel(), _ptr(count), _stream_ptr(dev)),
                       "pnr_render_touched")
        return index, count

    def clear_point_grads(self, embedding, color, direction, num_points: int, index: torch.Tensor,
                          count: Optional[torch.Tensor] = None) -> None:
        """pnr_point_grads_clear: zeroes the listed rows of dense point-gradient tensors (any may be None)."""
        dev = index.device
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_point_grads_clear(_ptr(embedding), _ptr(color), _ptr(direction), int(num_points),
                                                      _ptr(index), index.numel(), _ptr(count), _stream_ptr(dev)),
                       "pnr_point_grads_clear")

    PROBE_KEYS = {"ray_max_shading_opacity": ("d_max_opacity", ()), "ray_max_sample_loc_w": ("d_max_loc", (3,)),
                  "ray_max_far_dist": ("d_far_dist", ()), "shading_avg_color": ("d_avg_color", (3,)),
                  "shading_avg_dir": ("d_avg_dir", (3,)), "shading_avg_conf": ("d_avg_conf", ()),
                  "shading_avg_embedding": ("d_avg_embedding", (32,))}

    def probe(self) -> Dict[str, torch.Tensor]:
        """Probing outputs of the LAST render call (pnr_render_probe; the reference's legacy model with opt.prob == 1,
        models/neural_points_volumetric_model.py:331-352), under the reference's key names, over all R rays of the call
        (zeros for rays that are not kept) plus `ray_max_sample_index` (int32, -1 for those)."""
        if getattr(self, "_last", None) is None:
            raise RuntimeError("RendererHIP.probe: no render call to probe")
        d, R, arr, n, rc, rays_per_cam, cap = self._last
        if arr is None:
            raise RuntimeError("RendererHIP.probe: not after render_pose (render with host cameras to probe)")
        dev = self.scene.device
        out, pc = {}, _lib.ProbeC()
        for key, (field, tail) in self.PROBE_KEYS.items():
            out[key] = torch.empty((R,) + tail, dtype=torch.float32, device=dev)
            setattr(pc, field, out[key].data_ptr())
        out["ray_max_sample_index"] = torch.empty((R,), dtype=torch.int32, device=dev)
        pc.d_max_index = out["ray_max_sample_index"].data_ptr()
        ws = self._ws
        with torch.cuda.device(dev):
            _lib.check(self.lib.pnr_render_probe(self.scene.handle, arr, n, _ptr(rc), rays_per_cam, C.byref(self.opts), R,
                                                 _ptr(ws), ws.numel(), cap, C.byref(pc), _stream_ptr(dev)),
                       "pnr_render_probe")
        return out

    def touched_points(self) -> torch.Tensor:
        """Unique neighbour points of the last render call (int64, ascending): the rows of the point gradients that
        can be non-zero -- what a data-parallel step exchanges (distributed.GradExchange.reduce_points)."""
        if self._last is None:
            raise RuntimeError("RendererHIP.touched_points: no render call yet")
        R = self._last[1]
        taps = self.taps(R)
        n = int(taps["ray_off"][-1].item() + taps["ray_cnt"][-1].item())   # selected samples of the call
        pidx = taps["smp_pidx"].reshape(-1)[:min(n, self.cap_samples) * self.opts.K]
        return torch.unique(pidx[pidx >= 0]).to(torch.long)

    def taps(self, R: int):
        """Views into the last frame's workspace (tests): per selected sample loc+t, ray, pidx, decoded."""
        t = _lib.RenderTaps()
        ws = self._ws
        _lib.check(self.lib.pnr_render_taps(_ptr(ws), ws.numel(), R, self.cap_samples, self.opts.K, C.byref(t)),
                   "pnr_render_taps")
        base = ws.data_ptr()

        def view(ptr, nbytes, dtype, shape):
            off = ptr - base
            return ws[off:off + nbytes].view(dtype).view(*shape)
        cap, K = self.cap_samples, self.opts.K
        return {
            "smp_loc": view(t.smp_loc, cap * 16, torch.float32, (cap, 4)),
            "smp_ray": view(t.smp_ray, cap * 4, torch.int32, (cap,)),
            "smp_pidx": view(t.smp_pidx, cap * K * 4, torch.int32, (cap, K)),
            "smp_out": view(t.smp_out, cap * 16, torch.float32, (cap, 4)),
            "ray_cnt": view(t.ray_cnt, R * 4, torch.int32, (R,)),
            "ray_off": view(t.ray_off, R * 4, torch.int32, (R,)),
            "ray_dirs": view(t.ray_dirs, R * 12, torch.float32, (R, 3)),
        }
