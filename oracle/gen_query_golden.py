"""Generates tests/golden/query_stage.npz: drift-guard fixtures of the query stage.

TEST INFRASTRUCTURE.  The reference's query (`query_worldcoords.cu:18-433`) is CUDA-only and the reference holds no
fixture for it, so nothing here pins the CUDA binary.  What the fixtures do: they FREEZE the canonical sequential
semantics (voxel ids by first point, per-voxel lists in ascending point index
first P kept, the `voxel_idx > 0` drop of cu:147, layer -> x -> y -> z -> slot traversal and the replace-the-farthest
rule of cu:256-301) as int32 index lists, so that a later edit of oracle/query_oracle.c, of pnr_oracle.query_py or of
the HIP kernels cannot drift silently: all three are compared against these stored lists.

  python oracle/gen_query_golden.py      # needs neither the reference nor a GPU; rewrites tests/golden/query_stage.npz

Every case is produced by the C oracle AND asserted equal to the independent pure-Python statement (query_py)
before it is written.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pnr_oracle as O  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "query_stage.npz")

# (name, N, R, SR, K, P, compat)
CASES = [
    ("sr8_k8", 3000, 96, 8, 8, 12, True),
    ("sr80_k8", 3000, 64, 80, 8, 12, True),
    ("sr80_k12", 2500, 64, 80, 12, 26, True),
    ("sr8_k12_nocompat", 3000, 96, 8, 12, 8, False),
]


def scene(N, R, seed):
    """N points on a thin spherical cap of radius 0.12 around the origin (about one point per 0.008 voxel, some voxels
    well over P) and R rays from (0, 0.3, 1.1) towards points of a disc that is wider than the cap, so that some rays
    miss, some graze (hit the dilated occupancy, find no neighbour) and most cross both sides of the shell."""
    g = torch.Generator().manual_seed(seed)
    v = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    xyz = (v * (0.12 + (torch.rand(N, 1, generator=g) - 0.5) * 0.006)).float().contiguous()
    xyz[: N // 10] = xyz[N // 10: N // 10 + 1] + (torch.rand(N // 10, 3, generator=g) - 0.5) * 0.006  # a crowded voxel neighbourhood
    campos = torch.tensor([[0.0, 0.3, 1.1]])
    tgt = torch.cat([(torch.rand(R, 2, generator=g) - 0.5) * 0.36, torch.zeros(R, 1)], -1)
    dirs = torch.nn.functional.normalize(tgt - campos, dim=-1)[None].contiguous()
    return xyz, campos, dirs


def main():
    O.build_c_oracle()
    out = {}
    for i, (name, N, R, SR, K, P, compat) in enumerate(CASES):
        xyz, campos, dirs = scene(N, R, 40 + i)
        cfg = O.OracleConfig()
        cfg.SR, cfg.K, cfg.P, cfg.z_depth_dim = SR, K, P, 400
        cfg.ranges = [-0.3, -0.3, -0.3, 0.3, 0.3, 0.3]
        raypos, _ = O.ray_generation(campos, dirs, 400, 0.8, 1.6)
        ranges, svs, svd = O.get_hyperparameters(cfg, xyz)
        args = (raypos, xyz[None], cfg.kernel_size, cfg.query_size, SR, K, svd, cfg.max_o, P, O.radius_limit(cfg),
                ranges, svs, compat)
        pidx, loc, mask, stats = O.query(*args)
        p2, l2, m2 = O.query_py(*args)
        assert torch.equal(pidx, p2) and torch.equal(loc, l2) and torch.equal(mask, m2), name
        assert stats["rays_kept"] >= R // 3 and stats["rays_kept"] < R, (name, stats)
        assert stats["rays_hit"] > stats["rays_kept"], (name, stats)      # some hit rays find no neighbour
        full = (pidx >= 0).all(-1).sum().item()
        assert full >= 4, (name, full)                                     # samples whose K slots are all taken
        print(name, stats, "samples with K neighbours:", full)
        out.update({f"{name}_xyz": xyz.numpy(), f"{name}_campos": campos.numpy(), f"{name}_dirs": dirs.numpy(),
                    f"{name}_cfg": np.array([N, R, SR, K, P, int(compat), 400], dtype=np.int64),
                    f"{name}_nearfar": np.array([0.8, 1.6], dtype=np.float32),
                    f"{name}_ranges_cfg": np.array(cfg.ranges, dtype=np.float32),
                    f"{name}_pidx": pidx.numpy(), f"{name}_loc": loc.numpy(), f"{name}_mask": mask.numpy(),
                    f"{name}_stats": np.array([stats[k] for k in ("occupied_voxels", "rays_hit", "rays_kept",
                                                                  "valid_samples", "valid_pairs", "selected_samples")],
                                              dtype=np.int64)})
    out["names"] = np.array([c[0] for c in CASES])
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
