"""Seeded synthetic inputs of the shapes BASELINE.json names (datasets and checkpoints cannot be
downloaded): a neural point cloud on thin surfaces inside the nerf-synthetic "chair" bounding box,
nerf-synthetic style cameras, and Xavier-initialised MLP weights.

Conventions follow the reference: point tensor layouts of studio_utils.py:84-90, chair ranges of
dev_scripts/w_n360/chair_points.sh:57, cameras on a radius-4 sphere looking at the origin with
camera_angle_x = 0.6911 (focal 1111.1 px at W = 800), near 2 / far 6 (studio_datamanager.py:40-41),
unit-length ray directions as nerfstudio produces them.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

CHAIR_RANGES = [-0.721, -0.695, -0.995, 0.658, 0.706, 1.050]
LEGO_RANGES = [-0.638, -1.141, -0.346, 0.634, 1.149, 1.141]


def make_points(N: int, seed: int = 1234, ranges=CHAIR_RANGES, noise: float = 0.002) -> Dict[str, torch.Tensor]:
    """N points on a union of thin analytic surfaces (sphere shell, seat plate, back plate, four legs)
    with +-noise along the normal.  Returns the reference's state-dict layouts (CPU tensors)."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    lo = torch.tensor(ranges[:3])
    hi = torch.tensor(ranges[3:])
    n_sph = int(N * 0.35)
    n_seat = int(N * 0.25)
    n_back = int(N * 0.2)
    n_leg = N - n_sph - n_seat - n_back
    parts = []
    # sphere shell r = 0.42 around (0, 0, 0.25)
    v = torch.nn.functional.normalize(torch.randn(n_sph, 3, generator=g), dim=-1)
    parts.append(v * (0.42 + (rnd(n_sph, 1) - 0.5) * 2 * noise) + torch.tensor([0.0, 0.0, 0.25]))
    # seat plate z = -0.2
    p = torch.stack([(rnd(n_seat) - 0.5) * 1.1, (rnd(n_seat) - 0.5) * 1.1, -0.2 + (rnd(n_seat) - 0.5) * 2 * noise], -1)
    parts.append(p)
    # back plate x = -0.5
    p = torch.stack([-0.5 + (rnd(n_back) - 0.5) * 2 * noise, (rnd(n_back) - 0.5) * 1.1, -0.2 + rnd(n_back) * 1.1], -1)
    parts.append(p)
    # four legs: vertical cylinders r = 0.04
    ang = rnd(n_leg) * 2 * math.pi
    which = torch.randint(0, 4, (n_leg,), generator=g)
    cx = torch.tensor([-0.45, -0.45, 0.45, 0.45])[which]
    cy = torch.tensor([-0.45, 0.45, -0.45, 0.45])[which]
    rr = 0.04 + (rnd(n_leg) - 0.5) * 2 * noise
    p = torch.stack([cx + rr * torch.cos(ang), cy + rr * torch.sin(ang), -0.9 + rnd(n_leg) * 0.7], -1)
    parts.append(p)
    xyz = torch.cat(parts, 0)
    xyz = torch.max(torch.min(xyz, hi - 0.02), lo + 0.02)
    xyz = xyz[torch.randperm(N, generator=g)].float().contiguous()
    cam = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * 4.0
    return {
        "xyz": xyz,
        "embedding": (rnd(1, N, 32) - 0.5).float(),               # U(-0.5, 0.5), neural_points.py:285
        "conf": (0.1 + 0.9 * rnd(1, N, 1)).float(),
        "dir": torch.nn.functional.normalize(cam - xyz, dim=-1)[None].float().contiguous(),
        "color": rnd(1, N, 3).float(),
        "Rw2c": torch.eye(3),
    }


def make_room_points(N: int, seed: int = 4321, size=(8.0, 6.0, 3.0), noise: float = 0.004) -> Dict[str, torch.Tensor]:
    """N points on the six inner faces of a size[0] x size[1] x size[2] m room shell plus two box-shaped pieces of
    furniture (ScanNet-style indoor cloud: BASELINE.json configs[4], vsize 0.008, K = 12, near 0.1 / far 8,
    reference dev_scripts/w_scannet_etf/scene241_points.sh:53-60,91-92).  Same tensor layouts as make_points."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    sx, sy, sz = size
    n_wall = int(N * 0.8)
    face = torch.randint(0, 6, (n_wall,), generator=g)
    u, v = rnd(n_wall), rnd(n_wall)
    jit = (rnd(n_wall) - 0.5) * 2 * noise
    x = torch.where(face == 0, jit, torch.where(face == 1, sx + jit, u * sx))
    y = torch.where(face == 2, jit, torch.where(face == 3, sy + jit, torch.where(face < 2, u * sy, v * sy)))
    z = torch.where(face == 4, jit, torch.where(face == 5, sz + jit, v * sz))
    walls = torch.stack([x, y, z], -1)
    n_f = N - n_wall
    c = torch.tensor([[2.0, 2.0, 0.4], [5.5, 3.5, 0.5]])[torch.randint(0, 2, (n_f,), generator=g)]
    d = torch.nn.functional.normalize(torch.randn(n_f, 3, generator=g), dim=-1)
    d = d / d.abs().max(dim=-1, keepdim=True)[0]            # points on a cube surface
    furn = c + d * torch.tensor([0.6, 0.4, 0.4]) + (rnd(n_f, 3) - 0.5) * 2 * noise
    xyz = torch.cat([walls, furn], 0)[torch.randperm(N, generator=g)].float().contiguous()
    eye = torch.tensor([sx / 2, sy / 2, sz / 2])
    return {
        "xyz": xyz,
        "embedding": (rnd(1, N, 32) - 0.5).float(),
        "conf": (0.1 + 0.9 * rnd(1, N, 1)).float(),
        "dir": torch.nn.functional.normalize(eye - xyz, dim=-1)[None].float().contiguous(),
        "color": rnd(1, N, 3).float(),
        "Rw2c": torch.eye(3),
    }


def make_weights(seed: int = 0, sigma_scale: float = 1.0, bias_scale: float = 0.0) -> Dict[str, torch.Tensor]:
    """Xavier-uniform weights (reference models/helpers/networks.py:72-173: gain of leaky_relu(0.1) for hidden
    layers, 1 for the heads).  `sigma_scale` multiplies the density head so opacities are non-trivial with
    random weights; `bias_scale` > 0 gives U(-s, s) biases."""
    shapes = {
        "mlp_base.layers.0": (256, 284), "mlp_base.layers.1": (256, 256),
        "mlp_head.layers.0": (256, 263), "mlp_head.layers.1": (256, 256),
        "field_output_density.net": (1, 256),
        "mlp_color.layers.0": (128, 280), "mlp_color.layers.1": (128, 128), "mlp_color.layers.2": (128, 128),
        "field_output_color.net": (3, 128),
    }
    g = torch.Generator().manual_seed(seed)
    w = {}
    gain = torch.nn.init.calculate_gain("leaky_relu", 0.1)
    for name, (n_out, n_in) in shapes.items():
        last = name.startswith("field_output")
        std = (1.0 if last else gain) * np.sqrt(2.0 / (n_in + n_out))
        bound = float(std * np.sqrt(3.0))
        w[name + ".weight"] = (torch.rand((n_out, n_in), generator=g) * 2 - 1) * bound
        w[name + ".bias"] = (torch.rand((n_out,), generator=g) * 2 - 1) * bias_scale
    w["field_output_density.net.weight"] = w["field_output_density.net.weight"] * sigma_scale
    return w


def make_camera(azimuth_deg: float, elevation_deg: float = 30.0, radius: float = 4.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Camera on a sphere looking at the origin, OpenGL convention (camera looks down -z, +y up), as the
    blender/nerf-synthetic poses nerfstudio loads.  Returns (campos [3], camrotc2w [3,3])."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    pos = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az),
                        radius * math.sin(el)], dtype=torch.float64)
    back = pos / pos.norm()                       # +z of the camera points away from the scene
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(up, back)
    right = right / right.norm()
    true_up = torch.linalg.cross(back, right)
    rot = torch.stack([right, true_up, back], dim=1)   # columns = camera axes in world coordinates
    return pos.float(), rot.float().contiguous()


def make_inside_camera(pos, yaw_deg: float, pitch_deg: float = 0.0):
    """Camera at `pos` inside a scene (indoor configs), OpenGL convention, z up.  Returns (campos, camrotc2w)."""
    yaw, pitch = math.radians(yaw_deg), math.radians(pitch_deg)
    fwd = torch.tensor([math.cos(pitch) * math.cos(yaw), math.cos(pitch) * math.sin(yaw), math.sin(pitch)],
                       dtype=torch.float64)
    back = -fwd
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    right = torch.linalg.cross(up, back)
    right = right / right.norm()
    true_up = torch.linalg.cross(back, right)
    return torch.as_tensor(pos, dtype=torch.float32), torch.stack([right, true_up, back], dim=1).float().contiguous()


def make_rays(H: int, W: int, campos: torch.Tensor, camrot: torch.Tensor, camera_angle_x: float = 0.6911112070083618,
              y0: int = 0, y1: int = None, x0: int = 0, x1: int = None, full_W: int = None) -> torch.Tensor:
    """Unit ray directions [h*w, 3] of the pixel block [y0:y1, x0:x1] of an H x W pinhole image
    (pixel centres, nerfstudio convention)."""
    full_W = full_W or W
    focal = 0.5 * full_W / math.tan(0.5 * camera_angle_x)
    y1 = H if y1 is None else y1
    x1 = W if x1 is None else x1
    ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=torch.float32), torch.arange(x0, x1, dtype=torch.float32),
                            indexing="ij")
    d_cam = torch.stack([(xs + 0.5 - W * 0.5) / focal, -(ys + 0.5 - H * 0.5) / focal, -torch.ones_like(xs)], -1)
    d_world = d_cam.reshape(-1, 3) @ camrot.T
    return torch.nn.functional.normalize(d_world, dim=-1).contiguous()


# The configurations BASELINE.json names, as seeded synthetic stand-ins (datasets are not reachable).  `points`
# names the cloud generator, `camera` the camera family (orbit: nerf-synthetic sphere of radius 4; inside: a camera
# inside the cloud), `angle_x` the horizontal field of view (focal = 0.5 W / tan(0.5 angle_x)), `vsize` the voxel
# edge (the grid uses vsize * vscale, vscale = 2), near / far the clip planes of the datamanager.
SCENE_CONFIGS = {
    # configs[0]: 50k points, 64x64 image, 32 samples/ray, K = 8 (the reference's CPU-runnable case)
    "cfg0_chair_50k": dict(N=50_000, H=64, W=64, SR=32, K=8, ranges=CHAIR_RANGES, max_o=410000, P=12, vsize=0.004,
                           near=2.0, far=6.0, points="chair", camera="orbit", angle_x=0.6911112070083618),
    # configs[1]: ~6M points, 800x800, 80 samples/ray, K = 8 (the metric's configuration)
    "cfg1_chair_6m": dict(N=6_000_000, H=800, W=800, SR=80, K=8, ranges=CHAIR_RANGES, max_o=410000, P=12, vsize=0.004,
                          near=2.0, far=6.0, points="chair", camera="orbit", angle_x=0.6911112070083618),
    # configs[2]: lego-like, ~6M points, 800x800, K = 8 (the lego script's own numbers: bbox dev_scripts/w_n360/
    # lego_points.sh:59, max_o = 830000 :58, P = 9 :62)
    "cfg2_lego_6m": dict(N=6_000_000, H=800, W=800, SR=80, K=8, ranges=LEGO_RANGES, max_o=830000, P=9, vsize=0.004,
                         near=2.0, far=6.0, points="lego", camera="orbit", angle_x=0.6911112070083618),
    # configs[3]: DTU-like dense MVSNet cloud, ~10M points, 1600x1200 (W x H: a 4:3 frame), K = 8
    "cfg3_dtu_10m": dict(N=10_000_000, H=1200, W=1600, SR=80, K=8, ranges=CHAIR_RANGES, max_o=410000, P=12, vsize=0.004,
                         near=2.0, far=6.0, points="chair", camera="orbit", angle_x=0.9),
    # configs[4]: ScanNet-like indoor scene, ~20M points, 1296x968, K = 12, SR = 24, P = 26, vsize 0.008, the camera
    # inside the cloud (dev_scripts/w_scannet_etf/scene241_points.sh:53-60,91-92): the HBM-bound neighbour-gather stress
    "cfg4_scannet_20m": dict(N=20_000_000, H=968, W=1296, SR=24, K=12, ranges=[-0.5, -0.5, -0.5, 8.5, 6.5, 3.5],
                             max_o=1000000, P=26, vsize=0.008, near=0.1, far=8.0, points="room", camera="inside",
                             angle_x=1.0),
}


def make_scene_points(cfgd: dict, N: int = None, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """The cloud of a SCENE_CONFIGS entry (optionally at a reduced point count)."""
    n = int(N or cfgd["N"])
    if cfgd["points"] == "room":
        return make_room_points(n, seed=seed)
    if cfgd["points"] == "lego":
        # the chair-shaped surfaces squeezed into the lego box (flatter in x, longer in y)
        pts = make_points(n, seed=seed, ranges=CHAIR_RANGES)
        lo_c, hi_c = torch.tensor(CHAIR_RANGES[:3]), torch.tensor(CHAIR_RANGES[3:])
        lo_l, hi_l = torch.tensor(LEGO_RANGES[:3]), torch.tensor(LEGO_RANGES[3:])
        pts["xyz"] = ((pts["xyz"] - lo_c) / (hi_c - lo_c) * (hi_l - lo_l - 0.04) + lo_l + 0.02).float().contiguous()
        return pts
    return make_points(n, seed=seed, ranges=cfgd["ranges"])


def make_scene_camera(cfgd: dict, view: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Camera `view` (0..7) of a SCENE_CONFIGS entry: (campos [3], camrotc2w [3,3])."""
    if cfgd["camera"] == "inside":
        return make_inside_camera([4.0, 3.0, 1.5], yaw_deg=35.0 + 45.0 * view, pitch_deg=-10.0)
    return make_camera(45.0 * view + 20.0)
