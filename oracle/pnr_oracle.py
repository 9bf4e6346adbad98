"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Point-NeRF render hot path.

A plain PyTorch-CPU / numpy (+ one small C file for the index-heavy query stage)
restatement of the reference's per-ray hot path:

  Q  ray sampling + voxel query + fixed-K neighbour search
       pointnerf/models/rendering/diff_ray_marching.py:292-336   (ray_generation)
       pointnerf/nerfstudio/studio_utils.py:115-127              (get_hyperparameters)
       pointnerf/models/neural_points/cuda/query_worldcoords.cu:18-433 (query, in C)
  A  gather + dists + inverse-distance weights + PE + MLPs + K-aggregation
       pointnerf/nerfstudio/studio_utils.py:129-209              (neural_points_forward)
       pointnerf/nerfstudio/studio_model.py:263-365              (get_outputs, first half)
  C  ray_dist + front-to-back alpha composite + background fill
       pointnerf/nerfstudio/studio_model.py:368-399,491-504

Pinning (see tests/test_oracle_golden.py, oracle/gen_golden.py): stage A is checked
against the reference's own legacy ``PointAggregator`` (importable on CPU in the build
container), ray generation and the composite against the reference's
``near_far_linear_ray_generation`` / ``ray_march``.  The query stage (CUDA-only in the
reference, no reference test or fixture exists for it) is pinned only by this
restatement: **query parity is unpinned against the CUDA binary** (DESIGN.md).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package ``pointnerf2studio_amd`` never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------
# configuration: the PointNerfConfig fields the path reads (studio_model.py:61-118)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    num_viewdir_freqs: int = 4
    num_feat_freqs: int = 3
    num_dist_freqs: int = 5
    agg_dist_pers: int = 20
    point_features_dim: int = 32
    kernel_size: List[int] = field(default_factory=lambda: [3, 3, 3])
    vscale: List[int] = field(default_factory=lambda: [2, 2, 2])
    vsize: List[float] = field(default_factory=lambda: [0.004, 0.004, 0.004])
    query_size: List[int] = field(default_factory=lambda: [3, 3, 3])
    ranges: List[float] = field(default_factory=lambda: [-1.2, -1.2, -1.2, 1.2, 1.2, 1.2])
    z_depth_dim: int = 400
    SR: int = 80
    K: int = 8
    max_o: int = 1000000
    P: int = 12
    NN: int = 2
    gpu_maxthr: int = 1024


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def positional_encoding(x: torch.Tensor, num_freqs: int, ori: bool = False) -> torch.Tensor:
    """studio_utils.py:58-68 (identical to models/helpers/networks.py:176-191)."""
    freq_bands = (2 ** torch.arange(num_freqs).float()).to(x.device)
    ori_c = x.shape[-1]
    pts = (x[..., None] * freq_bands).reshape(x.shape[:-1] + (num_freqs * x.shape[-1],))
    if ori:
        pts = torch.cat([x, torch.sin(pts), torch.cos(pts)], dim=-1).reshape(
            pts.shape[:-1] + (pts.shape[-1] * 2 + ori_c,))
    else:
        pts = torch.stack([torch.sin(pts), torch.cos(pts)], dim=-1).reshape(
            pts.shape[:-1] + (pts.shape[-1] * 2,))
    return pts


def coarse_t_table(D: int, near: float, far: float) -> torch.Tensor:
    """Mid-point ray parameters of the D coarse samples at jitter 0.

    diff_ray_marching.py:307-323 evaluated for one ray: linspace -> near/far blend ->
    forward differences -> cumsum -> + near -> mid-points.  Returns float32 [D].
    (torch's CPU cumsum accumulates float32 inputs in double and rounds each prefix; the
    product path takes this table as an input, so both sides use the same numbers.)
    """
    tvals = torch.linspace(0, 1, D + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    seg = (tvals[..., 1:] - tvals[..., :-1]).view(1, 1, D)
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((1, 1, 1)), end], dim=2)
    end = near + end
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    return mid.reshape(D).contiguous()


def ray_generation(campos: torch.Tensor, raydir: torch.Tensor, point_count: int,
                   near: float, far: float, jitter: float = 0.0,
                   u: Optional[torch.Tensor] = None):
    """near_far_linear_ray_generation, diff_ray_marching.py:292-336.

    ``u`` replaces the reference's ``torch.rand((N, R, D))`` so a jittered run can be
    reproduced; with jitter == 0 it is unused.  Returns (raypos [N,R,D,3], t_mid [N,R,D]).
    """
    tvals = torch.linspace(0, 1, point_count + 1).view(1, -1)
    tvals = near * (1 - tvals) + far * tvals
    if u is None:
        u = torch.full((raydir.shape[0], raydir.shape[1], point_count), 0.5)
    seg = (tvals[..., 1:] - tvals[..., :-1]) * (1 + jitter * (u - 0.5))
    end = torch.cumsum(seg, dim=2)
    end = torch.cat([torch.zeros((end.shape[0], end.shape[1], 1)), end], dim=2)
    end = near + end
    mid = (end[:, :, :-1] + end[:, :, 1:]) / 2
    raypos = campos[:, None, None, :] + raydir[:, :, None, :] * mid[:, :, :, None]
    return raypos, mid


def jitter_uniforms(R: int, D: int, seed: int) -> torch.Tensor:
    """The counter-based uniforms the HIP path draws when jitter > 0 (pnr_uniform in
    pointnerf2studio_amd/csrc/pnr_internal.h, restated bit for bit): u[0, r, j] in [0, 1) with 24 random bits.
    They stand where the reference calls torch.rand((N, R, D)) (diff_ray_marching.py:316-319)."""
    m32 = np.uint64(0xFFFFFFFF)
    r = np.arange(R, dtype=np.uint64)[:, None]
    j = np.arange(D, dtype=np.uint64)[None, :]
    h = (np.uint64(seed & 0xFFFFFFFF) * np.uint64(0x9E3779B1) + r * np.uint64(0x85EBCA77) + j * np.uint64(0xC2B2AE3D)
         + np.uint64(0x27D4EB2F)) & m32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & m32
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & m32
    h ^= h >> np.uint64(15)
    u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy(u)[None]


def get_hyperparameters(cfg: OracleConfig, xyz: torch.Tensor):
    """studio_utils.py:104-127: voxel size, clipped + padded bbox, scaled grid dims.

    Returns (ranges float32[6] tensor, scaled_vsize float32[3] ndarray, scaled_vdim int32[3]).
    The numpy dtype promotions of the reference (python lists meeting float32 arrays) are
    kept: the padding and the grid dims are evaluated in float64.
    """
    vscale_np = np.array(cfg.vscale, dtype=np.int32)
    scaled_vsize_np = (cfg.vsize * vscale_np).astype(np.float32)          # :111
    min_xyz, max_xyz = torch.min(xyz, dim=-2)[0], torch.max(xyz, dim=-2)[0]
    rmin = torch.as_tensor(cfg.ranges[:3], dtype=torch.float32)
    rmax = torch.as_tensor(cfg.ranges[3:], dtype=torch.float32)
    min_xyz = torch.max(torch.stack([min_xyz, rmin], 0), 0)[0]
    max_xyz = torch.min(torch.stack([max_xyz, rmax], 0), 0)[0]
    pad = torch.as_tensor(scaled_vsize_np * cfg.kernel_size / 2, dtype=torch.float32)   # :121
    min_xyz = min_xyz - pad
    max_xyz = max_xyz + pad
    ranges = torch.cat([min_xyz, max_xyz], dim=-1)
    vdim = (max_xyz - min_xyz).numpy() / cfg.vsize                          # float64, :125
    scaled_vdim = np.ceil(vdim / vscale_np).astype(np.int32)
    return ranges, scaled_vsize_np, scaled_vdim


def radius_limit(cfg: OracleConfig) -> np.float32:
    """studio_utils.py:110."""
    return np.asarray(4 * max(cfg.vsize[0], cfg.vsize[1])).astype(np.float32)


# --------------------------------------------------------------------------------------
# query stage (C, sequential semantics)
# --------------------------------------------------------------------------------------
_LIB = None


def build_c_oracle(force: bool = False) -> str:
    so = os.path.join(_HERE, "libpnr_oracle.so")
    src = os.path.join(_HERE, "query_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpnr_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c_oracle())
        lib.pnr_oracle_query.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def query(raypos: torch.Tensor, xyz: torch.Tensor, kernel_size, query_size, SR: int, K: int,
          scaled_vdim, max_o: int, P: int, radius: float, ranges: torch.Tensor,
          scaled_vsize, compat_drop0: bool = True):
    """``woord_query_grid_point_index`` (query_worldcoords.cpp:33-78) on CPU, B = 1.

    raypos [1,R,D,3], xyz [1,N,3] (or [N,3]).  Returns
    (sample_pidx int32 [1,R'',SR,K], sample_loc f32 [1,R'',SR,3], ray_mask int8 [1,R], stats).
    """
    rp = np.ascontiguousarray(raypos.detach().numpy().reshape(raypos.shape[-3], raypos.shape[-2], 3), dtype=np.float32)
    R, D = rp.shape[0], rp.shape[1]
    pts = np.ascontiguousarray(xyz.detach().numpy().reshape(-1, 3), dtype=np.float32)
    N = pts.shape[0]
    ks = np.ascontiguousarray(kernel_size, dtype=np.int32)
    qs = np.ascontiguousarray(query_size, dtype=np.int32)
    dims = np.ascontiguousarray(scaled_vdim, dtype=np.int32)
    rg = np.ascontiguousarray(ranges.detach().numpy() if torch.is_tensor(ranges) else ranges, dtype=np.float32)
    vox = np.ascontiguousarray(scaled_vsize, dtype=np.float32)
    pidx = np.empty((max(R, 1), SR, K), dtype=np.int32)
    loc = np.empty((max(R, 1), SR, 3), dtype=np.float32)
    mask = np.zeros((max(R, 1),), dtype=np.int8)
    stats = np.zeros(8, dtype=np.int64)
    n = _lib().pnr_oracle_query(
        _p(rp), ctypes.c_int(R), ctypes.c_int(D), _p(pts), ctypes.c_int(N), _p(ks), _p(qs),
        ctypes.c_int(SR), ctypes.c_int(K), _p(dims), ctypes.c_int(max_o), ctypes.c_int(P),
        ctypes.c_float(float(radius)), _p(rg), _p(vox), ctypes.c_int(1 if compat_drop0 else 0),
        _p(pidx), _p(loc), _p(mask), _p(stats))
    if n < 0:
        raise RuntimeError("pnr_oracle_query failed")
    names = ["occupied_voxels", "max_o_overflow", "rays_hit", "rays_kept", "valid_samples",
             "valid_pairs", "selected_samples", "_"]
    return (torch.from_numpy(pidx[:n].copy())[None], torch.from_numpy(loc[:n].copy())[None],
            torch.from_numpy(mask[:R].copy())[None], dict(zip(names, stats.tolist())))


def query_py(raypos, xyz, kernel_size, query_size, SR, K, scaled_vdim, max_o, P, radius,
             ranges, scaled_vsize, compat_drop0=True):
    """Pure-Python statement of SURVEY.md Appendix A (small cases only): an independent
    second implementation used to cross-check the C oracle."""
    f32 = np.float32
    rp = raypos.detach().numpy().reshape(raypos.shape[-3], raypos.shape[-2], 3).astype(f32)
    pts = xyz.detach().numpy().reshape(-1, 3).astype(f32)
    shift = np.asarray(ranges, dtype=f32)[:3]
    vox = np.asarray(scaled_vsize, dtype=f32)
    dims = [int(v) for v in scaled_vdim]
    R, D = rp.shape[:2]

    def cell(p):
        c = [int(np.floor((f32(p[a]) - shift[a]) / vox[a])) for a in range(3)]
        ok = all(0 <= c[a] < dims[a] for a in range(3))
        return tuple(c), ok

    vid, coords, occ = {}, [], set()
    for i in range(pts.shape[0]):
        c, ok = cell(pts[i])
        if ok and c not in vid:
            vid[c] = len(coords)
            coords.append(c)
    for c in coords:
        for x in range(max(0, c[0] - query_size[0] // 2), min(dims[0], c[0] + (query_size[0] + 1) // 2)):
            for y in range(max(0, c[1] - query_size[1] // 2), min(dims[1], c[1] + (query_size[1] + 1) // 2)):
                for z in range(max(0, c[2] - query_size[2] // 2), min(dims[2], c[2] + (query_size[2] + 1) // 2)):
                    occ.add((x, y, z))
    cnt = [0] * len(coords)
    lst = [[] for _ in coords]
    for i in range(pts.shape[0]):
        c, ok = cell(pts[i])
        if not ok:
            continue
        v = vid[c]
        if (v > 0) if compat_drop0 else (v >= 0):
            if cnt[v] < P:
                lst[v].append(i)
            cnt[v] += 1
    r2 = f32(radius) * f32(radius)
    out_p, out_l, mask = [], [], np.zeros(R, dtype=np.int8)
    for r in range(R):
        cum, slots = 0, []
        for j in range(D):
            c, ok = cell(rp[r, j])
            m = 1 if (ok and c in occ) else 0
            cum += m
            if m and cum <= SR:
                slots.append(j)
        if cum == 0:
            continue
        pid = -np.ones((SR, K), dtype=np.int32)
        loc = np.zeros((SR, 3), dtype=f32)
        for s, j in enumerate(slots):
            loc[s] = rp[r, j]
            ctr = rp[r, j]
            (fx, fy, fz), _ = cell(ctr)
            kid, far2, far_ind = 0, f32(0), 0
            buf = np.zeros(K, dtype=f32)
            for layer in range((kernel_size[0] + 1) // 2):
                for x in range(max(-fx, -layer), min(dims[0] - fx, layer + 1)):
                    for y in range(max(-fy, -layer), min(dims[1] - fy, layer + 1)):
                        for z in range(max(-fz, -layer), min(dims[2] - fz, layer + 1)):
                            if max(abs(x), abs(y), abs(z)) != layer:
                                continue
                            v = vid.get((fx + x, fy + y, fz + z), -1)
                            if v < 0:
                                continue
                            for p in lst[v][:min(P, cnt[v])]:
                                d = pts[p] - ctr
                                d2 = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
                                if r2 == 0 or d2 <= r2:
                                    kid += 1
                                    if kid - 1 < K:
                                        pid[s, kid - 1] = p
                                        buf[kid - 1] = d2
                                        if d2 > far2:
                                            far2, far_ind = d2, kid - 1
                                    elif d2 < far2:
                                        pid[s, far_ind] = p
                                        buf[far_ind] = d2
                                        far2 = d2
                                        for i in range(K):
                                            if buf[i] > far2:
                                                far2, far_ind = buf[i], i
                if kid >= K:
                    break
        if (pid >= 0).any():
            mask[r] = 1
            out_p.append(pid)
            out_l.append(loc)
    n = len(out_p)
    pidx = np.stack(out_p) if n else np.zeros((0, SR, K), dtype=np.int32)
    locs = np.stack(out_l) if n else np.zeros((0, SR, 3), dtype=f32)
    return torch.from_numpy(pidx)[None], torch.from_numpy(locs)[None], torch.from_numpy(mask)[None]


# --------------------------------------------------------------------------------------
# NeuralPoints.forward (studio_utils.py:129-209)
# --------------------------------------------------------------------------------------
def w2pers(point_xyz, camrotc2w, campos):
    """studio_utils.py:129-135."""
    shift = point_xyz[None, ...] - campos[:, None, :]
    xyz = torch.sum(camrotc2w[:, None, :, :] * shift[:, :, :, None], dim=-2)
    xper = xyz[:, :, 0] / xyz[:, :, 2]
    yper = xyz[:, :, 1] / xyz[:, :, 2]
    return torch.stack([xper, yper, xyz[:, :, 2]], dim=-1)


def w2pers_loc(point_xyz_w, camrotc2w, campos):
    """studio_utils.py:137-144."""
    shift = point_xyz_w - campos[:, None, :]
    xyz_c = torch.sum(shift[..., None, :] * torch.transpose(camrotc2w, 1, 2)[:, None, None, ...], dim=-1)
    z = xyz_c[..., 2]
    return torch.stack([xyz_c[..., 0] / z, xyz_c[..., 1] / z, z], dim=-1)


def neural_points_forward(points: Dict[str, torch.Tensor], cfg: OracleConfig, origins, directions,
                          near: float, far: float, camrotc2w, jitter: float = 0.0,
                          u: Optional[torch.Tensor] = None, compat_drop0: bool = True, dtype=torch.float32):
    """studio_utils.py:147-209.  ``points`` holds xyz [N,3], embedding [1,N,32], conf [1,N,1],
    dir [1,N,3], color [1,N,3], Rw2c [3,3].  Returns the reference's 13-tuple plus (stats, t_mid
    of the selected samples [1,R'',SR] with 0 in unfilled slots).
    dtype: float32 is the reference's arithmetic.  float64 (tests only: the yardstick for how far ANY float32
    evaluation of the MLP chain sits from the exact value) keeps the float32 query -- same samples, same neighbours --
    and evaluates everything behind it in double."""
    cam_rot = camrotc2w.reshape(-1, 3, 3)[:1] if camrotc2w.shape[0] != 3 else camrotc2w[None]
    cam_rot = cam_rot.reshape(1, 3, 3).float()
    cam_pos = origins[0][None].float()
    ray_dirs = directions[None].float()
    xyz = points["xyz"].float()
    ranges, scaled_vsize, scaled_vdim = get_hyperparameters(cfg, xyz)
    raypos, t_mid = ray_generation(cam_pos, ray_dirs, cfg.z_depth_dim, near, far, jitter, u)
    pidx, loc_w, ray_mask, stats = query(
        raypos, xyz[None], cfg.kernel_size, cfg.query_size, cfg.SR, cfg.K, scaled_vdim, cfg.max_o,
        cfg.P, radius_limit(cfg), ranges, scaled_vsize, compat_drop0)
    keep = ray_mask[0] > 0
    if dtype != torch.float32:
        cam_rot, cam_pos, ray_dirs, xyz, loc_w = (t.to(dtype) for t in (cam_rot, cam_pos, ray_dirs, xyz, loc_w))
    sample_ray_dirs = ray_dirs[:, keep][..., None, :].expand(-1, -1, cfg.SR, -1).contiguous()
    pnt_mask = pidx >= 0
    B, R, SR, K = pidx.shape
    flat = torch.clamp(pidx, min=0).view(-1).long()
    loc = w2pers_loc(loc_w, cam_rot, cam_pos)
    pers = w2pers(xyz, cam_rot, cam_pos)
    emb = points["embedding"].to(dtype)
    table = torch.cat([xyz[None], pers, emb], dim=-1)
    sampled = torch.index_select(table, 1, flat).view(B, R, SR, K, emb.shape[2] + 6)
    s_color = torch.index_select(points["color"].to(dtype), 1, flat).view(B, R, SR, K, 3)
    s_dir = torch.index_select(points["dir"].to(dtype), 1, flat).view(B, R, SR, K, 3)
    s_conf = torch.index_select(points["conf"].to(dtype), 1, flat).view(B, R, SR, K, 1)
    return (s_color, points["Rw2c"].to(dtype), s_dir, sampled[..., 6:], sampled[..., 3:6], sampled[..., :3], s_conf,
            loc, loc_w, pnt_mask, sample_ray_dirs, cfg.vsize, ray_mask), stats


# --------------------------------------------------------------------------------------
# get_outputs (studio_model.py:263-399) -- nerfstudio MLP / FieldHead / RGBRenderer restated
# --------------------------------------------------------------------------------------
MLP_SHAPES = {
    "mlp_base.layers.0": (256, 284), "mlp_base.layers.1": (256, 256),
    "mlp_head.layers.0": (256, 263), "mlp_head.layers.1": (256, 256),
    "field_output_density.net": (1, 256),
    "mlp_color.layers.0": (128, 280), "mlp_color.layers.1": (128, 128), "mlp_color.layers.2": (128, 128),
    "field_output_color.net": (3, 128),
}


def _mlp(x, w, name, n_layers, slope=0.1):
    """nerfstudio MLP with activation == out_activation == LeakyReLU(0.1) (studio_model.py:193-216):
    a Linear followed by the activation, n_layers times."""
    for i in range(n_layers):
        x = F.leaky_relu(F.linear(x, w[f"{name}.layers.{i}.weight"], w[f"{name}.layers.{i}.bias"]), slope)
    return x


def linear_weights(dists, pnt_mask):
    """PointNerf.linear with axis_weight == [1,1,1] (studio_model.py:467-475)."""
    return pnt_mask * (1.0 / torch.clamp(torch.norm(dists[..., :3], dim=-1), min=1e-6))


def conf_coefficient(s_conf):
    """studio_model.py:288-292 (== the legacy gradiant_clamp, point_aggregators.py:816-818): clamp(conf, 1e-4, 1) with
    a straight-through gradient.  s_conf [..., K, 1] -> [..., K]."""
    conf = s_conf[..., 0]
    return conf - (conf - torch.clamp(conf, min=0.0001, max=1)).detach()


def decode_features(w: Dict[str, torch.Tensor], cfg: OracleConfig, s_color, Rw2c, s_dir, s_emb, s_pers, s_xyz,
                    loc, loc_w, pnt_mask, sample_ray_dirs, s_conf=None, slope: float = 0.1):
    """studio_model.py:270-365: per-sample [sigma, r, g, b]; zero where no neighbour.  Returns (decoded, sample_valid,
    weight): `weight` is the normalised inverse-distance weight the legacy forward also hands back
    (point_aggregators.py:813-830).  The plugin never multiplies by the confidence (s_conf=None); with s_conf the
    LEGACY path runs: the aggregation uses weight x clamp(conf) (point_aggregators.py:816-826) -- what the probing
    outputs average with.  slope: LeakyReLU slope (0.1 in the plugin, studio_model.py:197-198; the legacy net was
    trained with 0.01)."""
    sample_valid = torch.any(pnt_mask, dim=-1).view(-1)
    total_len = len(sample_valid)
    in_shape = loc_w.shape
    xdist = s_pers[..., 0] * s_pers[..., 2] - loc[:, :, :, None, 0] * loc[:, :, :, None, 2]
    ydist = s_pers[..., 1] * s_pers[..., 2] - loc[:, :, :, None, 1] * loc[:, :, :, None, 2]
    zdist = s_pers[..., 2] - loc[:, :, :, None, 2]
    dists = torch.stack([xdist, ydist, zdist], dim=-1)
    dists = torch.cat([s_xyz - loc_w[..., None, :], dists], dim=-1)
    weight = linear_weights(dists, pnt_mask)
    weight = weight / torch.clamp(torch.sum(weight, dim=-1, keepdim=True), min=1e-8)

    flat = pnt_mask.view(-1)
    viewdirs = sample_ray_dirs.reshape(-1, 3)
    B, R, SR, K, _ = dists.shape
    Rt = Rw2c.transpose(-1, -2)
    viewdirs = viewdirs @ Rt
    viewdirs = positional_encoding(viewdirs, cfg.num_viewdir_freqs, ori=True)
    ori_view, viewdirs = viewdirs[..., :3], viewdirs[..., 3:]
    viewdirs = viewdirs[sample_valid, :]

    d = dists.view(-1, 6)[flat, :]
    d[..., :3] = d[..., :3] @ Rt
    d = positional_encoding(d, cfg.num_dist_freqs)
    feat = s_emb.reshape(-1, s_emb.shape[-1])[flat, :]
    feat = torch.cat([feat, positional_encoding(feat, cfg.num_feat_freqs)], dim=-1)
    feat = torch.cat([feat, d], dim=-1)
    norm_weight = weight
    if s_conf is not None:
        weight = weight * conf_coefficient(s_conf)
    weight = weight.view(B * R * SR, K, 1)
    feat = _mlp(feat, w, "mlp_base", 2, slope)

    col = s_color.reshape(-1, 3)[flat, :]
    feat = torch.cat([feat, col], dim=-1)
    sdir = s_dir.reshape(-1, 3)[flat, :] @ Rt
    ov = ori_view[..., None, :].repeat(1, K, 1).view(-1, 3)[flat, :]
    feat = torch.cat([feat, sdir - ov, torch.sum(sdir * ov, dim=-1, keepdim=True)], dim=-1)
    feat = _mlp(feat, w, "mlp_head", 2, slope)

    alpha = F.relu(F.linear(feat, w["field_output_density.net.weight"], w["field_output_density.net.bias"]))
    holder = torch.zeros([B * R * SR * K, 1], dtype=feat.dtype)
    holder[flat, :] = alpha
    alpha = torch.sum(holder.view(B * R * SR, K, 1) * weight, dim=-2).view(-1, 1)[sample_valid, :]
    holder = torch.zeros([B * R * SR * K, feat.shape[-1]], dtype=feat.dtype)
    holder[flat, :] = feat
    agg = torch.sum(holder.view(B * R * SR, K, -1) * weight, dim=-2).view(-1, feat.shape[-1])[sample_valid, :]

    c = torch.cat([agg, viewdirs], dim=-1)
    c = _mlp(c, w, "mlp_color", 3, slope)
    c = torch.sigmoid(F.linear(c, w["field_output_color.net.weight"], w["field_output_color.net.bias"]))
    c = c * (1 + 2 * 0.001) - 0.001
    out = torch.zeros([total_len, 4], dtype=c.dtype)
    out[sample_valid] = torch.cat([alpha, c], dim=-1)
    return out.view(in_shape[:-1] + (4,)), sample_valid.view(in_shape[:-1]), norm_weight.view(B, R, SR, K)


def compute_ray_dist(loc, sample_valid, vsize):
    """studio_model.py:368-375: per-sample segment length from the running max of the
    camera-space z of the SR slots (unfilled slots hold the camera-space image of the world
    origin -- reproduced, not fixed); last slot and out-of-range gaps fall back to vsize[2]."""
    ray_dist = torch.cummax(loc[..., 2], dim=-1)[0]
    ray_dist = torch.cat([ray_dist[..., 1:] - ray_dist[..., :-1],
                          torch.full((ray_dist.shape[0], ray_dist.shape[1], 1), vsize[2])], dim=-1)
    mask = ray_dist < 1e-8
    mask = torch.logical_or(mask, ray_dist > 2 * vsize[2]).to(torch.float32)
    ray_dist = ray_dist * (1.0 - mask) + mask * vsize[2]
    return ray_dist * sample_valid.float()


def alpha_composite(decoded, sample_valid, ray_dist):
    """studio_model.py:379-386 == ray_march, diff_ray_marching.py:495-541: opacity, exclusive
    cumprod transmittance, blend weights and the un-backgrounded colour sum."""
    sigma = decoded[..., 0] * sample_valid.float()
    opacity = 1 - torch.exp(-sigma * ray_dist)
    acc_t = torch.cumprod(1. - opacity + 1e-10, dim=-1)
    bg_t = acc_t[:, :, [-1]]
    acc_t = torch.cat([torch.ones(opacity.shape[0:2] + (1,)), acc_t[:, :, :-1]], dim=-1)
    bw = (opacity * acc_t).unsqueeze(-1)
    comp = torch.sum(bw * decoded[..., 1:4], dim=-2)
    return comp, opacity, acc_t, bw, bg_t


def composite(decoded, sample_valid, loc, vsize, training: bool = False, ts=None):
    """studio_model.py:368-390: ray_dist, alpha composite, white-background RGBRenderer
    (``comp + bg * (1 - sum w)``; eval-time clamp to [0,1] as nerfstudio's RGBRenderer does
    outside training [ns-mem]).  Returns rgb [B,R,3], acc [B,R], depth [B,R] (the build's own
    definition: sum(w*t)/(sum(w)+1e-6), formula of models/neural_points_volumetric_model.py:319-322),
    blend weights [B,R,SR]."""
    ray_dist = compute_ray_dist(loc, sample_valid, vsize)
    comp, opacity, acc_t, bw, _ = alpha_composite(decoded, sample_valid, ray_dist)
    acc = torch.sum(bw, dim=-2)
    rgb = comp + torch.ones(3) * (1.0 - acc)
    if not training:
        rgb = torch.clamp(rgb, min=0.0, max=1.0)
    depth = None
    if ts is not None:
        depth = (bw[..., 0] * ts).sum(-1) / (bw[..., 0].sum(-1) + 1e-6)
    return rgb, acc[..., 0], depth, bw[..., 0]


def fill_invalid(rgb_hit, ray_mask):
    """studio_model.py:491-504: scatter hit rays into a white [R,3] image."""
    B, OR = ray_mask.shape
    inds = torch.nonzero(ray_mask)
    out = torch.ones([B, OR, 3], dtype=rgb_hit.dtype)
    out[inds[..., 0], inds[..., 1], :] = rgb_hit.reshape(-1, 3)
    return out.squeeze(0)


def render(points, w, cfg: OracleConfig, origins, directions, near, far, camrotc2w,
           jitter: float = 0.0, u=None, training: bool = False, compat_drop0: bool = True, probe: bool = False,
           dtype=torch.float32):
    """NeuralPoints.forward + PointNerf.get_outputs for one ray bundle.  Returns a dict with
    the plugin's outputs (coarse_raycolor [R,3], ray_mask [R] int8) plus the build's extra
    outputs (depth [R], acc [R]) and intermediate tensors used by the parity tests."""
    (s_color, Rw2c, s_dir, s_emb, s_pers, s_xyz, s_conf, loc, loc_w, pnt_mask, ray_dirs, vsize,
     ray_mask), stats = neural_points_forward(points, cfg, origins, directions, near, far, camrotc2w,
                                              jitter, u, compat_drop0, dtype)
    if dtype != torch.float32:
        # the yardstick (tests): the per-sample decode in `dtype` behind the float32 query; no composite
        w = {k: v.to(dtype) for k, v in w.items()}
        decoded, sample_valid, weight = decode_features(w, cfg, s_color, Rw2c, s_dir, s_emb, s_pers, s_xyz, loc, loc_w,
                                                        pnt_mask, ray_dirs)
        return {"ray_mask": ray_mask.squeeze(0), "stats": stats, "decoded": decoded, "sample_valid": sample_valid,
                "pnt_mask": pnt_mask}
    R = directions.shape[0]
    out = {"ray_mask": ray_mask.squeeze(0), "stats": stats}
    if loc_w.shape[1] == 0:
        out.update(coarse_raycolor=torch.ones(R, 3), depth=torch.zeros(R), acc=torch.zeros(R))
        return out
    decoded, sample_valid, weight = decode_features(w, cfg, s_color, Rw2c, s_dir, s_emb, s_pers, s_xyz, loc, loc_w,
                                                    pnt_mask, ray_dirs)
    # ray parameter of each selected sample (for the depth output): t = <loc_w - o, d> / <d, d>
    o = origins[0][None, None, None, :].float()
    ts = torch.sum((loc_w - o) * ray_dirs, dim=-1) / torch.sum(ray_dirs * ray_dirs, dim=-1)
    rgb_hit, acc_hit, depth_hit, bw = composite(decoded, sample_valid, loc, vsize, training, ts)
    out["coarse_raycolor"] = fill_invalid(rgb_hit, ray_mask)
    keep = ray_mask[0] > 0
    depth = torch.zeros(R)
    acc = torch.zeros(R)
    depth[keep] = depth_hit[0]
    acc[keep] = acc_hit[0]
    out.update(depth=depth, acc=acc, decoded=decoded, sample_valid=sample_valid, sample_pidx=None,
               sample_loc_w=loc_w, blend_weight=bw, pnt_mask=pnt_mask, agg_weight=weight)
    if training:
        out["conf_coefficient"] = conf_coefficient(s_conf)      # studio_model.py:288-292,396-397
    if probe:
        out["probe"] = probe_outputs(decoded, sample_valid, loc, loc_w, vsize, weight, s_conf, s_xyz, s_color, s_dir,
                                     s_emb, pnt_mask, ray_mask, R)
    return out


def get_loss_dict(outputs, image, training: bool = True, zero_epsilon: float = 1e-3,
                  zero_one_loss_weights: float = 1e-4) -> Dict[str, torch.Tensor]:
    """PointNerf.get_loss_dict (studio_model.py:415-431; config defaults :117-118): MSELoss (mean over the elements of
    the kept rays) + 1e-6, and in training mean(log v + log(1 - v)) * zero_one_loss_weights over the clamped
    conf_coefficient tensor.  loss_coefficients is nerfstudio's default (every key 1.0) [ns-mem]."""
    keep = (outputs["ray_mask"] > 0)[..., None].expand(-1, 3)
    masked_output = torch.masked_select(outputs["coarse_raycolor"], keep).reshape(-1, 3)
    masked_gt = torch.masked_select(image, keep).reshape(-1, 3)
    loss = {"ray_masked_coarse_raycolor_loss": torch.mean((masked_gt - masked_output) ** 2) + 1e-6}
    if training:
        val = torch.clamp(outputs["conf_coefficient"], zero_epsilon, 1 - zero_epsilon)
        loss["conf_coefficient_loss"] = torch.mean(torch.log(val) + torch.log(1 - val)) * zero_one_loss_weights
    return loss


# --------------------------------------------------------------------------------------
# probing outputs + point grow / prune
# --------------------------------------------------------------------------------------
def probe_outputs(decoded, sample_valid, loc, loc_w, vsize, weight, s_conf, s_xyz, s_color, s_dir, s_emb, pnt_mask,
                  ray_mask, R):
    """models/neural_points_volumetric_model.py:331-352 (`opt.prob == 1`): per ray, the shading sample of largest
    opacity (`coarse_point_opacity` = 1 - exp(-sigma * ray_dist), first maximum), its world position, the distance of
    its nearest neighbour and the K-averages of its neighbours' colour / dir / conf / embedding with the weights the
    legacy aggregator returns in probe mode: normalised inverse-distance weight x clamp(conf, 1e-4, 1)
    (models/aggregators/point_aggregators.py:816-830).  Opacity uses this build's ray_dist (the plugin's,
    studio_model.py:368-375).  One stated deviation: `ray_max_far_dist` takes the minimum over the FILLED neighbour
    slots (the legacy gather reads point 0 through unfilled slots, neural_points.py clamp(pidx, 0)); 1e10 if none.
    Rays that are not kept: all zeros, max_index -1.  Returns tensors over ALL R rays."""
    ray_dist = compute_ray_dist(loc, sample_valid, vsize)
    _, opacity, _, _, _ = alpha_composite(decoded, sample_valid, ray_dist)          # [1,R'',SR]
    max_op, ind = torch.max(opacity, dim=-1, keepdim=True)
    # torch.max does not promise WHICH maximum on ties; the canonical choice is the first
    first = (opacity == max_op).float().argmax(dim=-1, keepdim=True)
    ind = first
    g3 = lambda t: torch.gather(t, 2, ind[..., None].expand(-1, -1, -1, t.shape[-1])).squeeze(2)
    g4 = lambda t: torch.gather(t, 2, ind[..., None, None].expand(-1, -1, -1, t.shape[-2], t.shape[-1])).squeeze(2)
    max_loc = g3(loc_w)                                                                  # [1,R'',3]
    wk = g3(weight * conf_coefficient(s_conf))[..., None]                                                  # [1,R'',K,1]
    nb_xyz, nb_mask = g4(s_xyz), g3(pnt_mask.float()) > 0
    d = torch.norm(nb_xyz - max_loc[..., None, :], dim=-1)
    d = torch.where(nb_mask, d, torch.full_like(d, 1e10))
    far = torch.min(d, dim=-1)[0]
    vals = dict(max_opacity=max_op[0, :, 0], max_loc=max_loc[0], far_dist=far[0],
                avg_color=torch.sum(g4(s_color) * wk, dim=-2)[0], avg_dir=torch.sum(g4(s_dir) * wk, dim=-2)[0],
                avg_conf=torch.sum(g4(s_conf) * wk, dim=-2)[0, :, 0], avg_embedding=torch.sum(g4(s_emb) * wk, dim=-2)[0],
                max_index=ind[0, :, 0].to(torch.int32))
    keep = ray_mask[0] > 0
    full = {}
    for k, v in vals.items():
        t = torch.full((R,) + tuple(v.shape[1:]), -1 if k == "max_index" else 0, dtype=v.dtype)
        t[keep] = v
        full[k] = t
    return full


def prune_points(points: Dict[str, torch.Tensor], thresh: float):
    """models/neural_points/neural_points.py:341-364: keep the points whose confidence is >= thresh.  Returns the new
    tensors and old_index [N'] (the index each kept point had)."""
    mask = points["conf"][0, ..., 0] >= thresh
    out = {"xyz": points["xyz"][mask, :], "embedding": points["embedding"][:, mask, :],
           "conf": points["conf"][:, mask, :], "dir": points["dir"][:, mask, :], "color": points["color"][:, mask, :],
           "Rw2c": points["Rw2c"]}
    return out, torch.nonzero(mask).reshape(-1).to(torch.int32)


def grow_points(points: Dict[str, torch.Tensor], add_xyz, add_embedding, add_color, add_dir, add_conf):
    """models/neural_points/neural_points.py:367-393: the added points are appended behind the existing ones."""
    N = points["xyz"].shape[0]
    out = {"xyz": torch.cat([points["xyz"], add_xyz], dim=0),
           "embedding": torch.cat([points["embedding"], add_embedding[None, ...]], dim=1),
           "conf": torch.cat([points["conf"], add_conf[None, ...]], dim=1),
           "dir": torch.cat([points["dir"], add_dir[None, ...]], dim=1),
           "color": torch.cat([points["color"], add_color[None, ...]], dim=1), "Rw2c": points["Rw2c"]}
    old = torch.cat([torch.arange(N, dtype=torch.int32), torch.full((add_xyz.shape[0],), -1, dtype=torch.int32)])
    return out, old


# --------------------------------------------------------------------------------------
# seeded weights (Xavier-uniform per models/helpers/networks.py:72-173, zero bias)
# --------------------------------------------------------------------------------------
def make_weights(seed: int = 0, sigma_scale: float = 1.0, bias_scale: float = 0.0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w = {}
    gain = torch.nn.init.calculate_gain("leaky_relu", 0.1)
    for name, (n_out, n_in) in MLP_SHAPES.items():
        last = name.startswith("field_output")
        std = (1.0 if last else gain) * np.sqrt(2.0 / (n_in + n_out))
        bound = float(std * np.sqrt(3.0))
        w[name + ".weight"] = (torch.rand((n_out, n_in), generator=g) * 2 - 1) * bound
        w[name + ".bias"] = (torch.rand((n_out,), generator=g) * 2 - 1) * bias_scale
    w["field_output_density.net.weight"] = w["field_output_density.net.weight"] * sigma_scale
    return w
