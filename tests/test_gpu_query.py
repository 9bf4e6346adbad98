"""GPU parity of the drop-in query op (pnr_query_raypos behind woord_query_grid_point_index) against the
sequential CPU oracle of query_worldcoords.cu: neighbour index lists BIT-EXACT, sample locations exact,
ray mask exact."""
import numpy as np
import pytest
import torch

from helpers import build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd.renderer import query_raypos

pytestmark = pytest.mark.gpu


def _run_case(oracle, device, N, SR, K, P, compat, H=40, W=40, az=35.0, shrink=1.0, seed=1234, ks=3, D=400):
    pts = small_scene(N, seed=seed, shrink=shrink)
    cfg = oracle_cfg(oracle, SR=SR, K=K, P=P, D=D)
    cfg.kernel_size = [ks] * 3
    cfg.query_size = [ks] * 3
    campos, camrot, dirs = camera_rays(H, W, az=az)
    raypos, _ = oracle.ray_generation(campos[None], dirs[None], cfg.z_depth_dim, 2.0, 6.0)
    ranges, svsize, svdim = oracle.get_hyperparameters(cfg, pts["xyz"])
    ref_pidx, ref_loc, ref_mask, stats = oracle.query(
        raypos, pts["xyz"][None], cfg.kernel_size, cfg.query_size, SR, K, svdim, cfg.max_o, P,
        oracle.radius_limit(cfg), ranges, svsize, compat)
    scene, _, hyp, info = build_hip(pts, cfg, device, compat=compat)
    assert np.array_equal(hyp.scaled_vdim, svdim) and np.array_equal(hyp.ranges, ranges.numpy())
    assert info["occupied_voxels"] == stats["occupied_voxels"]
    pidx, loc, mask, cnt = query_raypos(scene, raypos.to(device), SR, K, float(oracle.radius_limit(cfg)))
    assert cnt["rays_hit"] == stats["rays_hit"]
    assert cnt["rays_kept"] == stats["rays_kept"]
    assert torch.equal(mask.cpu(), ref_mask), "ray_mask differs"
    assert pidx.shape == ref_pidx.shape
    assert torch.equal(pidx.cpu(), ref_pidx), "neighbour index lists are not bit-exact"
    assert torch.equal(loc.cpu(), ref_loc), "sample locations differ"
    assert cnt["pairs_valid"] >= stats["valid_pairs"]  # GPU counts pairs of rays the post-filter drops too
    return stats


@pytest.mark.parametrize("N,SR,K,P,compat", [
    (30000, 80, 8, 12, True),     # sparse cloud: K rarely full, both search layers visited
    (400000, 80, 8, 12, True),    # dense cloud: P cap active, replace-the-farthest rule exercised
    (400000, 8, 8, 12, True),     # SR cap active (more hits than slots)
    (200000, 24, 12, 26, True),   # ScanNet-style K = 12 (undefined in the reference: KN = 8), P = 26
    (200000, 80, 8, 9, False),    # lego-style P = 9, without the voxel-0 compat drop
    (50000, 32, 8, 12, True),     # BASELINE.json configs[0] sizes
])
def test_query_bit_exact(oracle, gpu_device, N, SR, K, P, compat):
    stats = _run_case(oracle, gpu_device, N, SR, K, P, compat)
    assert stats["rays_kept"] > 50 and stats["valid_pairs"] > 1000


@pytest.mark.parametrize("N,SR,K,P,ks,D", [
    (120000, 40, 8, 12, 5, 400),    # 5x5x5 search: three layers, the generic (cell by cell) search kernel
    (120000, 40, 20, 12, 5, 400),   # ... with K = 20 (> 16: the widest register variant)
    (60000, 80, 32, 26, 3, 400),    # K = 32 = PNR_MAX_K on the batched kernel's fallback
    (60000, 80, 8, 12, 1, 400),     # 1x1x1 search: only the sample's own voxel
    (90000, 16, 3, 5, 3, 130),      # D not a multiple of 64, K = 3
    (90000, 5, 1, 1, 3, 64),        # one neighbour, one point per voxel, one occupancy word
])
def test_query_bit_exact_corner_shapes(oracle, gpu_device, N, SR, K, P, ks, D):
    stats = _run_case(oracle, gpu_device, N, SR, K, P, True, H=24, W=24, ks=ks, D=D)
    assert stats["rays_kept"] > 20 and stats["valid_pairs"] > 50


def test_query_dense_shrunk_cloud(oracle, gpu_device):
    # 300k points squeezed into 35 % of the extent: ~30 points per voxel >> P, every list capped
    stats = _run_case(oracle, gpu_device, 300000, 80, 8, 12, True, shrink=0.35, H=32, W=32)
    assert stats["valid_pairs"] > 1000


def test_query_no_hits(oracle, gpu_device):
    # camera looking away from the cloud: no ray hits, empty outputs (edge case of cu:386)
    pts = small_scene(20000)
    cfg = oracle_cfg(oracle)
    scene, _, hyp, _ = build_hip(pts, cfg, gpu_device)
    campos = torch.tensor([[0.0, 0.0, 4.0]])
    dirs = torch.nn.functional.normalize(torch.tensor([[[0.0, 0.1, 1.0], [0.1, 0.0, 1.0]]]), dim=-1)
    raypos, _ = oracle.ray_generation(campos, dirs, 400, 2.0, 6.0)
    pidx, loc, mask, cnt = query_raypos(scene, raypos.to(gpu_device), 80, 8, 0.016)
    assert pidx.shape == (1, 0, 80, 8) and loc.shape == (1, 0, 80, 3)
    assert mask.sum().item() == 0 and cnt["rays_hit"] == 0


def test_scene_build_is_deterministic(oracle, gpu_device):
    # the structure is built with atomics; its CONTENT must not depend on their order
    pts = small_scene(300000, shrink=0.5)
    cfg = oracle_cfg(oracle)
    campos, camrot, dirs = camera_rays(24, 24)
    raypos, _ = oracle.ray_generation(campos[None], dirs[None], 400, 2.0, 6.0)
    outs = []
    for _ in range(3):
        scene, _, _, _ = build_hip(pts, cfg, gpu_device)
        pidx, loc, mask, _ = query_raypos(scene, raypos.to(gpu_device), 80, 8, 0.016)
        outs.append((pidx.cpu(), mask.cpu()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])


def _hip_query(args, device):
    """The HIP op on the argument tuple of oracle.query: (raypos, xyz, ks, qs, SR, K, vdim, max_o, P, radius, ranges,
    vsize, compat)."""
    from pointnerf2studio_amd.renderer import SceneHIP
    raypos, xyz, ks, qs, SR, K, vdim, max_o, P, radius, ranges, vsize, compat = args
    scene = SceneHIP()
    info = scene.build(xyz.reshape(-1, 3).to(device), np.asarray(ranges, dtype=np.float32), vsize, vdim, ks, qs, P,
                       max_o, compat)
    pidx, loc, mask, cnt = query_raypos(scene, raypos.to(device), SR, K, float(radius))
    return pidx.cpu(), loc.cpu(), mask.cpu(), cnt, info


def test_query_matches_stored_fixtures(oracle, gpu_device):
    """pnr_query_raypos against the int32 lists frozen in tests/golden/query_stage.npz (SR in {8, 80}, K in {8, 12},
    compat on / off): the HIP kernels, the C oracle and the Python statement are all held to the same stored lists
    (the CPU half is tests/test_query_fixtures.py)."""
    from query_cases import stored_case_args, stored_cases
    g, names = stored_cases()
    for name in names:
        args, want, stats = stored_case_args(oracle, g, name)
        pidx, loc, mask, cnt, info = _hip_query(args, gpu_device)
        assert torch.equal(mask, want[2]), name
        assert torch.equal(pidx, want[0]), f"{name}: neighbour lists differ from the stored fixture"
        assert torch.equal(loc, want[1]), name
        assert [info["occupied_voxels"], cnt["rays_hit"], cnt["rays_kept"]] == stats[:3].tolist()
        assert cnt["samples_selected"] >= stats[5]   # the GPU counts the samples of hit rays the post-filter drops too


@pytest.mark.parametrize("compat", [True, False])
def test_query_hand_derived_case(gpu_device, compat):
    """Expected lists derived by hand from query_worldcoords.cu (tests/query_cases.py): more than P points in a voxel,
    a replace-the-farthest step with a tie, the voxel-0 drop, a hit ray without neighbours, a partial list."""
    from query_cases import hand_case
    args, want, stats = hand_case(compat)
    pidx, loc, mask, cnt, info = _hip_query(args, gpu_device)
    assert torch.equal(mask, want[2])
    assert torch.equal(pidx, want[0]), pidx
    assert torch.equal(loc, want[1])
    assert cnt["rays_hit"] == stats["rays_hit"] and cnt["rays_kept"] == stats["rays_kept"]
    assert info["occupied_voxels"] == stats["occupied_voxels"]
