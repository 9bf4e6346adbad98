"""Shared by the CPU and GPU query tests: the stored drift-guard fixtures (tests/golden/query_stage.npz, generator
oracle/gen_query_golden.py) and one HAND-DERIVED case whose expected lists are worked out below from the reference's
CUDA source alone (query_worldcoords.cu), not from any implementation in this repository."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def stored_cases():
    g = np.load(os.path.join(GOLD, "query_stage.npz"))
    return g, [str(n) for n in g["names"]]


def stored_case_args(oracle, g, name):
    """(args of oracle.query / query_py, expected pidx, loc, mask, stats) of a stored case."""
    N, R, SR, K, P, compat, D = [int(v) for v in g[f"{name}_cfg"]]
    xyz = torch.from_numpy(g[f"{name}_xyz"])
    campos, dirs = torch.from_numpy(g[f"{name}_campos"]), torch.from_numpy(g[f"{name}_dirs"])
    near, far = [float(v) for v in g[f"{name}_nearfar"]]
    cfg = oracle.OracleConfig()
    cfg.SR, cfg.K, cfg.P, cfg.z_depth_dim = SR, K, P, D
    cfg.ranges = [float(v) for v in g[f"{name}_ranges_cfg"]]
    raypos, _ = oracle.ray_generation(campos, dirs, D, near, far)
    ranges, svs, svd = oracle.get_hyperparameters(cfg, xyz)
    args = (raypos, xyz[None], cfg.kernel_size, cfg.query_size, SR, K, svd, cfg.max_o, P, oracle.radius_limit(cfg),
            ranges, svs, bool(compat))
    want = (torch.from_numpy(g[f"{name}_pidx"]), torch.from_numpy(g[f"{name}_loc"]), torch.from_numpy(g[f"{name}_mask"]))
    return args, want, g[f"{name}_stats"]


# ----------------------------------------------------------------------------------------------------------------
# Hand-derived case.  Grid: origin (0,0,0), voxel 1.0, 4 x 4 x 4 cells, kernel_size = query_size = 3 (one ring of
# neighbours, two search layers), SR = 2, K = 2, P = 3, radius_limit = 1.5 (r^2 = 2.25), D = 4 explicit positions.
#
# points (index: position -> cell)                                voxel ids in order of first point (cu:49-73,
#   0: (1.5, 1.5, 1.5) -> (1,1,1)                                  sequential semantics): (1,1,1) = 0, (2,1,1) = 1,
#   1: (2.2, 1.5, 1.5) -> (2,1,1)                                  (3,3,3) = 2
#   2: (2.8, 1.5, 1.5) -> (2,1,1)
#   3: (2.5, 1.5, 1.5) -> (2,1,1)          per-voxel lists, first P = 3 in point order (cu:117-162):
#   4: (2.5, 1.6, 1.5) -> (2,1,1)            voxel 1: [1, 2, 3]   (point 4 is the 4th of its voxel: not kept)
#   5: (1.4, 1.5, 1.5) -> (1,1,1)            voxel 2: [6]
#   6: (3.5, 3.5, 3.5) -> (3,3,3)            voxel 0: [0, 5] -- but `voxel_idx > 0` (cu:147) drops voxel 0's points
#                                                     when compat is on
# occupancy (cu:80-115): every cell within one cell of ANY claimed voxel (voxel 0 included):
#   x,y,z in [0,2]^3  U  [1,3]x[0,2]x[0,2]  U  [2,3]^3
#
# ray 0: j0 (5,5,5) outside the grid; j1 (2.5,1.5,1.5) cell (2,1,1) occupied -> slot 0; j2 (0.5,1.5,1.5) cell
#        (0,1,1) occupied (ring of (1,1,1)) -> slot 1; j3 (2.6,1.5,1.5) occupied but a third hit > SR: dropped.
#   slot 0, centre (2.5,1.5,1.5), layer 0 = its own cell = voxel 1, candidates in list order (cu:262-296):
#     p1: d2 = 0.3^2 = 0.09 <= 2.25 -> out[0] = 1, far2 = 0.09, far_ind = 0
#     p2: d2 = 0.3^2 (same float: 2.2f and 2.8f are symmetric about 2.5) -> out[1] = 2; not > far2: far_ind stays 0
#     p3: d2 = 0 -> third candidate, K = 2 taken, 0 < far2 -> replaces out[far_ind = 0] = 3; far2 rescan -> slot 1
#     end of layer 0 with kid = 3 >= K: stop (cu:300).                                   => [3, 2]
#   slot 1, centre (0.5,1.5,1.5), cell (0,1,1): layer 0 empty; layer 1 reaches x in {0,1}: only (1,1,1) = voxel 0.
#     compat on : voxel 0 holds no points                                                => [-1, -1]
#     compat off: p0 d2 = 1.0 -> out[0]; p5 d2 = 0.9^2 = 0.81 -> out[1]                 => [0, 5]
# ray 1: all four positions in cell (0,3,0): not occupied -> not hit.
# ray 2: j0 (3.5,0.5,2.5) cell (3,0,2): occupied (ring of (2,1,1)), slot 0; layer 1 reaches voxel 1 = cell (2,1,1):
#        d2 = 3.69 / 2.49 / 3.0 for p1 / p2 / p3, all > 2.25 -> no neighbour: the ray is hit but NOT kept (cu:425-429).
# ray 3: j0 (3.5,3.4,3.5) cell (3,3,3) = voxel 2, layer 0: p6 d2 = 0.01 -> out[0] = 6, kid = 1 < K -> layer 1: nothing.
#                                                                                        => [6, -1]
# kept rays in order: 0, 3.  Unfilled slots: pidx -1, loc (0,0,0) (cu:383-384).
# ----------------------------------------------------------------------------------------------------------------
def hand_case(compat: bool):
    xyz = torch.tensor([[1.5, 1.5, 1.5], [2.2, 1.5, 1.5], [2.8, 1.5, 1.5], [2.5, 1.5, 1.5], [2.5, 1.6, 1.5],
                        [1.4, 1.5, 1.5], [3.5, 3.5, 3.5]], dtype=torch.float32)
    out_of_grid = [5.0, 5.0, 5.0]
    raypos = torch.tensor([
        [out_of_grid, [2.5, 1.5, 1.5], [0.5, 1.5, 1.5], [2.6, 1.5, 1.5]],
        [[0.5, 3.5, 0.5], [0.6, 3.5, 0.5], [0.7, 3.5, 0.5], [0.8, 3.5, 0.5]],
        [[3.5, 0.5, 2.5], out_of_grid, out_of_grid, out_of_grid],
        [[3.5, 3.4, 3.5], out_of_grid, out_of_grid, out_of_grid],
    ], dtype=torch.float32)[None]
    kw = dict(kernel_size=[3, 3, 3], query_size=[3, 3, 3], SR=2, K=2, scaled_vdim=np.array([4, 4, 4], dtype=np.int32),
              max_o=100, P=3, radius=1.5, ranges=torch.tensor([0.0, 0.0, 0.0, 4.0, 4.0, 4.0]),
              scaled_vsize=np.array([1.0, 1.0, 1.0], dtype=np.float32))
    args = (raypos, xyz[None], kw["kernel_size"], kw["query_size"], kw["SR"], kw["K"], kw["scaled_vdim"], kw["max_o"],
            kw["P"], kw["radius"], kw["ranges"], kw["scaled_vsize"], compat)
    slot1 = [-1, -1] if compat else [0, 5]
    pidx = torch.tensor([[[3, 2], slot1], [[6, -1], [-1, -1]]], dtype=torch.int32)[None]
    loc = torch.tensor([[[2.5, 1.5, 1.5], [0.5, 1.5, 1.5]], [[3.5, 3.4, 3.5], [0.0, 0.0, 0.0]]], dtype=torch.float32)[None]
    mask = torch.tensor([[1, 0, 0, 1]], dtype=torch.int8)
    return args, (pidx, loc, mask), dict(rays_hit=3, rays_kept=2, occupied_voxels=3)
