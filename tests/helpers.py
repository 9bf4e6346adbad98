"""Shared helpers of the parity tests: a small seeded scene + the oracle/HIP plumbing."""
import numpy as np
import torch

from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import (RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters)


def small_scene(N=60000, seed=1234, shrink=1.0):
    pts = synthetic.make_points(N, seed=seed)
    if shrink != 1.0:
        pts["xyz"] = (pts["xyz"] * shrink).contiguous()
    return pts


def oracle_cfg(oracle, SR=80, K=8, P=12, ranges=None, max_o=410000, D=400):
    cfg = oracle.OracleConfig()
    cfg.SR, cfg.K, cfg.P, cfg.max_o, cfg.z_depth_dim = SR, K, P, max_o, D
    cfg.ranges = list(ranges or synthetic.CHAIR_RANGES)
    return cfg


def build_hip(points, cfg, device, weights=None, compat=True):
    xyz = points["xyz"].to(device)
    hyp = grid_hyperparameters(xyz, cfg.vsize, cfg.vscale, cfg.kernel_size, cfg.ranges)
    scene = SceneHIP()
    info = scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, cfg.kernel_size, cfg.query_size, cfg.P,
                       cfg.max_o, compat)
    scene.pack_points(xyz, points["embedding"].to(device), points["conf"].to(device), points["dir"].to(device),
                      points["color"].to(device))
    wh = None
    if weights is not None:
        wh = WeightsHIP()
        wh.pack(weights, points["Rw2c"], device)
    return scene, wh, hyp, info


def camera_rays(H, W, az=35.0, el=30.0, window=None):
    campos, camrot = synthetic.make_camera(az, el)
    if window is None:
        dirs = synthetic.make_rays(H, W, campos, camrot)
    else:
        y0, y1, x0, x1 = window
        dirs = synthetic.make_rays(H, W, campos, camrot, y0=y0, y1=y1, x0=x0, x1=x1)
    return campos, camrot, dirs
