"""C-ABI and host-side logic that must work WITHOUT a GPU: the library loads and exports every symbol
include/pnr.h declares, argument validation fails loudly, the host mirrors of the reference's hyper-parameter
code agree with the oracle, and the product package never reaches into oracle/."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pointnerf2studio_amd import build, _lib
    build.build_library()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from pointnerf2studio_amd import _lib
    header = open(os.path.join(ROOT, "include", "pnr.h")).read()
    declared = set(re.findall(r"\b(pnr_[a-z0-9_]+)\s*\(", header))
    declared -= {"pnr_scene", "pnr_weights"}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.pnr_version() == 100


def test_struct_layouts_match_header(lib):
    from pointnerf2studio_amd import _lib
    assert C.sizeof(_lib.GridParams) == 6 * 4 + 3 * 4 + 3 * 4 + 3 * 4 + 3 * 4 + 3 * 4
    assert C.sizeof(_lib.CameraC) == (3 + 9 + 2) * 4
    assert C.sizeof(_lib.RenderOpts) == 13 * 4 + 4 + 8 + 8        # + pad, d_tape, tape_bytes
    assert C.sizeof(_lib.ViewC) == (3 + 9 + 2 + 4) * 4
    # ... and against the library itself (sizeof as compiled: pnr_abi_sizes)
    got = (C.c_int64 * 8)()
    assert lib.pnr_abi_sizes(C.byref(got)) == 0
    want = [C.sizeof(_lib.GridParams), C.sizeof(_lib.CameraC), C.sizeof(_lib.RenderOpts), C.sizeof(_lib.ViewC),
            C.sizeof(_lib.GradsC), C.sizeof(_lib.ProbeC), C.sizeof(_lib.RenderTaps), _lib.RenderOpts.d_tape.offset]
    assert list(got) == want


def test_pinhole_ray_host_statement(lib):
    """pnr_pinhole_ray is the host statement of the ray arithmetic pnr_render_camera runs in its kernels (nerfstudio's
    pinhole convention: pixel centres, camera looking down -z, unit directions; studio_datamanager.py:62-110 hands
    such bundles to the model).  Against the numpy formula in float64 and against synthetic.make_rays (torch)."""
    from pointnerf2studio_amd import synthetic
    from pointnerf2studio_amd.renderer import View, pinhole_ray
    campos, camrot = synthetic.make_camera(35.0, 20.0)
    H, W = 1200, 1600
    v = View(campos, camrot, fx=1650.0, fy=1660.5, cx=790.25, cy=611.5)
    R = camrot.double().numpy()
    for (x, y) in [(0, 0), (1599, 1199), (800, 600), (13, 977), (1234, 5)]:
        dc = np.array([(x + 0.5 - 790.25) / 1650.0, -(y + 0.5 - 611.5) / 1660.5, -1.0])
        want = R @ dc
        want /= np.linalg.norm(want)
        got = pinhole_ray(v, x, y)
        assert got.dtype == np.float32 and abs(np.linalg.norm(got.astype(np.float64)) - 1.0) < 1e-6
        assert np.abs(got - want).max() < 2e-7
    v2 = View.from_angle(campos, camrot, H, W, 0.9)
    d = synthetic.make_rays(H, W, campos, camrot, 0.9, y0=300, y1=302, x0=40, x1=43).numpy()
    for i, (y, x) in enumerate([(300, 40), (300, 41), (300, 42), (301, 40), (301, 41), (301, 42)]):
        assert np.abs(pinhole_ray(v2, x, y) - d[i]).max() < 2e-7


def test_argument_validation_fails_loudly(lib):
    from pointnerf2studio_amd import _lib
    rc = lib.pnr_scene_build(None, None, 10, None, None)
    assert rc == -1 and b"null" in lib.pnr_last_error()
    with pytest.raises(RuntimeError, match="null"):
        _lib.check(rc, "pnr_scene_build")
    assert lib.pnr_render(None, None, None, 1, None, None, None, None, None, None, None, None, None, 0, 1, None) == -1
    assert lib.pnr_query_raypos(None, None, 1, 400, 80, 8, 0.016, None, None, None, None, None, 0, None) == -1
    assert lib.pnr_render_camera(None, None, None, 1, 8, 8, None, 64, None, None, None, None, None, None, None, None, 0, 1,
                                 None) == -1
    view = (_lib.ViewC * 1)()
    assert lib.pnr_camera_rays(view, 1, 8, 8, None, 64, None, None) == -1 and b"focal" in lib.pnr_last_error()
    view[0].fx = view[0].fy = 10.0
    assert lib.pnr_camera_rays(view, 1, 8, 8, None, 65, None, None) == -1 and b"n_pixels" in lib.pnr_last_error()
    assert lib.pnr_camera_rays(view, 17, 8, 8, None, 64, None, None) == -1 and b"n_views" in lib.pnr_last_error()


def test_workspace_sizes_are_monotonic(lib):
    a = lib.pnr_render_workspace_bytes(1000, 10000, 8)
    b = lib.pnr_render_workspace_bytes(1000, 20000, 8)
    c = lib.pnr_render_workspace_bytes(2000, 20000, 8)
    d = lib.pnr_render_workspace_bytes(2000, 20000, 12)
    assert 0 < a < b < c < d
    # the aggregated-feature buffer (1 KiB per sample) dominates
    assert b - a >= 10000 * 1024
    q = lib.pnr_query_workspace_bytes(1000, 400, 80, 8)
    assert 0 < q < lib.pnr_render_workspace_bytes(1000, 80000, 8)
    # the scene-dependent size (bf16x3 point table) needs a built scene: fails loudly without one
    assert lib.pnr_render_workspace_bytes_for(None, None, 1000, 10000) == 0
    assert b"pnr_render_workspace_bytes_for" in lib.pnr_last_error()


def test_host_tensors_are_rejected_not_silently_computed():
    """No CPU fallback: handing the product path a CPU tensor raises."""
    from pointnerf2studio_amd.renderer import SceneHIP
    with pytest.raises(RuntimeError, match="GPU"):
        SceneHIP().build(torch.zeros(10, 3), np.zeros(6), np.ones(3), [4, 4, 4], [3, 3, 3], [3, 3, 3], 12, 100)


def test_missing_library_fails_loudly(monkeypatch):
    from pointnerf2studio_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpnr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_grid_hyperparameters_mirror_reference(oracle):
    from pointnerf2studio_amd.renderer import coarse_t_table, grid_hyperparameters
    from pointnerf2studio_amd import synthetic
    for seed, ranges in [(1, synthetic.CHAIR_RANGES), (2, [-1.2] * 3 + [1.2] * 3), (3, [-0.3, -0.3, -0.3, 0.2, 0.4, 0.3])]:
        xyz = synthetic.make_points(5000, seed=seed)["xyz"]
        cfg = oracle.OracleConfig()
        cfg.ranges = list(ranges)
        r, s, d = oracle.get_hyperparameters(cfg, xyz)
        h = grid_hyperparameters(xyz, cfg.vsize, cfg.vscale, cfg.kernel_size, ranges)
        assert np.array_equal(h.ranges, r.numpy()) and np.array_equal(h.scaled_vsize, s)
        assert np.array_equal(h.scaled_vdim, d)
    for D, n, f in [(400, 2.0, 6.0), (400, 0.1, 8.0), (64, 2.0, 6.0)]:
        assert torch.equal(coarse_t_table(D, n, f), oracle.coarse_t_table(D, n, f))


def test_synthetic_weights_match_oracle_generator(oracle):
    from pointnerf2studio_amd import synthetic
    a, b = synthetic.make_weights(0, 40.0, 0.1), oracle.make_weights(0, 40.0, 0.1)
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert {k: tuple(v.shape) for k, v in a.items() if k.endswith("weight")} == \
        {k + ".weight": v for k, v in oracle.MLP_SHAPES.items()}


def test_synthetic_scene_and_rays():
    from pointnerf2studio_amd import synthetic
    p = synthetic.make_points(20000, seed=5)
    lo, hi = torch.tensor(synthetic.CHAIR_RANGES[:3]), torch.tensor(synthetic.CHAIR_RANGES[3:])
    assert p["xyz"].shape == (20000, 3) and torch.all(p["xyz"] >= lo) and torch.all(p["xyz"] <= hi)
    assert p["embedding"].shape == (1, 20000, 32) and p["conf"].shape == (1, 20000, 1)
    assert torch.allclose(p["dir"].norm(dim=-1), torch.ones(1, 20000), atol=1e-5)
    campos, rot = synthetic.make_camera(30.0)
    assert torch.allclose(rot.T @ rot, torch.eye(3), atol=1e-6) and abs(campos.norm().item() - 4.0) < 1e-5
    d = synthetic.make_rays(8, 8, campos, rot)
    assert d.shape == (64, 3) and torch.allclose(d.norm(dim=-1), torch.ones(64), atol=1e-6)
    centre = synthetic.make_rays(800, 800, campos, rot, y0=400, y1=401, x0=400, x1=401)
    assert torch.allclose(centre[0], -campos / campos.norm(), atol=2e-3)   # the camera looks at the origin
    win = synthetic.make_rays(800, 800, campos, rot, y0=10, y1=12, x0=20, x1=23)
    full = synthetic.make_rays(800, 800, campos, rot).view(800, 800, 3)
    assert torch.allclose(win.view(2, 3, 3), full[10:12, 20:23], atol=1e-6)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pointnerf2studio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pnr_oracle" not in src and "oracle/" not in src and "import oracle" not in src, f
                assert "/root/reference" not in src, f


def test_jitter_uniform_matches_oracle_generator(lib, oracle):
    """The host-callable copy of the kernels' counter-based uniform against the oracle's numpy restatement."""
    u = oracle.jitter_uniforms(37, 400, seed=7)[0].numpy()
    for r, j in [(0, 0), (0, 399), (5, 17), (36, 1), (36, 399), (12, 200)]:
        assert lib.pnr_jitter_uniform(7, r, j) == float(u[r, j])
    u2 = oracle.jitter_uniforms(4, 16, seed=0xFFFFFFFF)[0].numpy()
    assert lib.pnr_jitter_uniform(0xFFFFFFFF, 3, 15) == float(u2[3, 15])
    big = oracle.jitter_uniforms(2000, 400, seed=3)[0].numpy()
    assert 0.0 <= big.min() and big.max() < 1.0 and abs(big.mean() - 0.5) < 2e-3
    assert abs(np.corrcoef(big[:, :-1].ravel(), big[:, 1:].ravel())[0, 1]) < 5e-3


def test_backward_workspace_size_and_grads_struct(lib):
    """Host-only entry points of the training step: the workspace size is a pure function of (cap_samples, K), grows
    with both, and pnr_grads_t is 21 device pointers (3 point tensors + 9 weights + 9 biases) + the sparse triple
    (rows, indices, capacity)."""
    from pointnerf2studio_amd import _lib
    a = lib.pnr_backward_workspace_bytes(4096, 8)
    b = lib.pnr_backward_workspace_bytes(8192, 8)
    c = lib.pnr_backward_workspace_bytes(4096, 12)
    assert 0 < a < b and a < c
    # ~10 KB per (sample, slot) row: the tapes X0 288 + H1 256 + H2 264 + G1 256 + G2 256 floats, the gradients at the four
    # pre-activations (4 x 256 floats: the exact mode's chain kernel writes them for the weight GEMMs), the row gradients
    # (40) and the mask bits (128 B)
    per_row = (b - a) / (4096 * 8)
    assert 9500 < per_row < 11500, per_row
    assert lib.pnr_backward_workspace_bytes(0, 0) == lib.pnr_backward_workspace_bytes(1, 1)   # clamped, never 0
    assert C.sizeof(_lib.GradsC) == 24 * C.sizeof(C.c_void_p)


def test_row_sparse_adam_entry_points_validate_arguments(lib):
    """pnr_rows_merge / pnr_adam_rows (the optimiser half of the training step): host-side validation, struct layout."""
    from pointnerf2studio_amd import _lib
    assert C.sizeof(_lib.AdamTensorC) == 4 * C.sizeof(C.c_void_p) + 8     # four pointers + width, padded to 8
    assert lib.pnr_rows_merge(None, 10, None, None, 10, None, 4, None, None) == -1 and b"null" in lib.pnr_last_error()
    one = (_lib.AdamTensorC * 1)()
    assert lib.pnr_adam_rows(one, 0, 10, None, 10, None, 0.9, 0.999, 1e-8, 1e-3, 1.0, None) == -1
    assert b"n_tensors" in lib.pnr_last_error()
    assert lib.pnr_adam_rows(one, 1, 10, None, 5, None, 0.9, 0.999, 1e-8, 1e-3, 1.0, None) == -1
    assert b"every row" in lib.pnr_last_error()
    assert lib.pnr_adam_rows(one, 1, 10, None, 10, None, 0.9, 0.999, 1e-8, 1e-3, 1.0, None) == -1
    assert b"null pointer" in lib.pnr_last_error()
    assert lib.pnr_adam_rows(one, 1, 10, None, 10, None, 0.4, 0.999, 1e-8, 1e-3, 1.0, None) == -1
    assert b"beta1" in lib.pnr_last_error()
    from pointnerf2studio_amd.optim import PointRowAdam
    with pytest.raises(ValueError, match="weight decay"):
        PointRowAdam([torch.nn.Parameter(torch.zeros(1, 4, 3))], weight_decay=0.1)
    opt = PointRowAdam([torch.nn.Parameter(torch.zeros(1, 4, 3))], lr=2e-3)
    opt.param_groups[0]["params"][0].grad = torch.zeros(1, 4, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step()


def test_row_publication_reaches_point_row_adam_parameters_only():
    """publish_rows (called by the fused backward after every step) hands the row list to the PointRowAdam that owns a
    parameter, which merges it into its ever-touched set at once: nothing is stored, so nothing can accumulate -- index
    tensors are up to tens of MB -- under another optimiser, without optimiser steps, or after the optimiser is gone."""
    import gc
    from pointnerf2studio_amd import optim
    a, b = torch.nn.Parameter(torch.zeros(1, 8, 3)), torch.nn.Parameter(torch.zeros(1, 8, 3))
    idx = torch.arange(4, dtype=torch.int32)
    optim.publish_rows([a, b, None], idx, None)                            # nobody owns them: a no-op
    assert a not in optim._SUBSCRIBED and b not in optim._SUBSCRIBED
    opt = optim.PointRowAdam([a], lr=1e-3)
    assert optim._SUBSCRIBED[a]() is opt and b not in optim._SUBSCRIBED
    for _ in range(100):
        optim.publish_rows([a, b], idx, None)                              # (host tensors: nothing to merge into, no state kept)
    assert not hasattr(optim, "_PENDING") and len(opt._listed) == 0
    del opt
    gc.collect()
    assert optim._SUBSCRIBED[a]() is None
    optim.publish_rows([a], idx, None)                                     # the owner is gone: a no-op again
