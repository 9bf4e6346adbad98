"""GPU parity of pnr_render_backward (the gradients of a render) against torch
autograd through the CPU oracle of NeuralPoints.forward + PointNerf.get_outputs (studio_model.py:263-399,
studio_utils.py:190-209) -- the very computation `ns-train pointnerf-original` differentiates.

Tolerance: fp32 mode -- every gradient tensor within 2e-3 of its own largest magnitude (fp32 MFMA products summed
in a different order than torch's CPU GEMMs, float atomics; measured ~1e-6); bf16x3 mode -- see BF16X3_* below; the
recomputed image within 1e-4 abs in both."""
import pytest
import torch

from helpers import NORTH_STAR, OPT_IN_BF16X3, build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER, RendererHIP

pytestmark = pytest.mark.gpu

GRAD_REL_TOL = NORTH_STAR["grad_rel"]   # the default (fp32) mode's bar


def _oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, training, jitter=0.0, u=None):
    pts_g = dict(pts)
    for k in ("embedding", "color", "dir"):
        pts_g[k] = pts[k].clone().requires_grad_(True)
    w_g = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    ref = oracle.render(pts_g, w_g, cfg, campos[None].expand(dirs.shape[0], 3), dirs, 2.0, 6.0, camrot,
                        jitter=jitter, u=u, training=training)
    loss = (ref["coarse_raycolor"] * G).sum()
    loss.backward()
    grads = {k: pts_g[k].grad.reshape(pts_g[k].shape[-2], -1) for k in ("embedding", "color", "dir")}
    for k, v in w_g.items():
        grads[k] = v.grad if v.grad is not None else torch.zeros_like(v)
    return ref, grads


# bf16x3 backward: the MLP forward is recomputed with 2^-16-relative products, so a few hundred of the ~70 M
# pre-activations that lie within ~1e-5 of zero land on the other side of the LeakyReLU kink than in the fp32 oracle;
# each such unit changes its own gradient contribution by 0.9x.  The gradient is exact for the function the bf16x3
# mode computes; against the fp32 oracle it shows as ~sqrt(flipped fraction) = a few 1e-3 in relative L2 and up to a
# few 1e-2 of the largest magnitude on single entries (tensors behind few kinks -- colour MLP, heads -- agree to 1e-5).
BF16X3_L2_TOL = OPT_IN_BF16X3["grad_l2"]
BF16X3_MAX_TOL = OPT_IN_BF16X3["grad_max"]


def _compare(name, got, want, bf16x3=False):
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert scale > 0, f"{name}: the oracle gradient is identically zero (test scene too empty)"
    if bf16x3:
        l2 = (got - want).norm().item() / want.norm().item()
        assert l2 <= BF16X3_L2_TOL and err <= BF16X3_MAX_TOL * scale, \
            f"{name}: rel L2 {l2:.3e}, max abs err {err:.3e} vs scale {scale:.3e}"
        return l2
    assert err <= GRAD_REL_TOL * scale, f"{name}: max abs err {err:.3e} vs scale {scale:.3e} ({err / scale:.2e} rel)"
    return err / scale


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])   # arithmetic of the backward's forward / data GEMMs
@pytest.mark.parametrize("N,SR,K,P,H,W,az,training", [
    (60000, 80, 8, 12, 24, 24, 35.0, True),      # training composite (no clamp)
    (50000, 32, 8, 12, 32, 32, 200.0, False),    # eval clamp: gradient passes only inside [0, 1]
    (200000, 24, 12, 26, 20, 20, 300.0, True),   # K = 12
    (60000, 32, 4, 12, 24, 24, 120.0, True),     # K = 4: tape rows = 4 per sample (tiles of 32 rows hold 8 samples)
    (250000, 16, 32, 26, 14, 14, 60.0, True),    # K = PNR_MAX_K = 32: a sample fills a 32-row tile (generic kernels)
    (250000, 16, 17, 26, 14, 14, 150.0, False),  # K = 17: the first K past the 16-lane segment
])
def test_backward_matches_oracle_autograd(oracle, gpu_device, N, SR, K, P, H, W, az, training, precision):
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=SR, K=K, P=P)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(H, W, az=az)
    torch.manual_seed(5)
    G = torch.randn(dirs.shape[0], 3)
    ref, want = _oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, training)
    assert ref["acc"].max().item() > 0.5

    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                      vsize_z=cfg.vsize[2], precision=precision, eval_clamp=not training)
    out = rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    assert out["counters"]["overflow"] == 0
    got = rnd.backward(G.to(gpu_device), w, pts["xyz"].shape[0])

    err = (got["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item()
    assert err <= 1e-4, f"recomputed image differs from the oracle by {err:.3e}"
    err2 = (got["rgb"] - out["rgb"]).abs().max().item()
    assert err2 <= 1e-4, f"recomputed image differs from pnr_render by {err2:.3e}"
    rel = {}
    bf = precision == "bf16x3"
    for k in ("embedding", "color", "dir"):
        rel[k] = _compare(k, got[k].cpu(), want[k], bf)
    for name in MLP_TENSOR_ORDER:
        for suf in (".weight", ".bias"):
            rel[name + suf] = _compare(name + suf, got[name + suf].cpu(), want[name + suf], bf)
    if bf:   # few kinks between these tensors and the loss: plain arithmetic agreement
        for name in ("mlp_color.layers.0", "mlp_color.layers.2", "field_output_color.net", "field_output_density.net"):
            assert rel[name + ".weight"] <= 2e-4, (name, rel[name + ".weight"])
    print("relative gradient errors:", {k: f"{v:.1e}" for k, v in rel.items()})


def test_backward_after_bf16x3_render_and_accumulation(oracle, gpu_device):
    """After a render in the opt-in bf16x3 mode the backward runs its forward / data GEMMs in bf16x3 too, and a
    second backward call with other cotangents returns the gradients of THAT call (buffers are fresh zeros)."""
    N, SR, K, P = 60000, 80, 8, 12
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=SR, K=K, P=P)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(24, 24, az=35.0)
    torch.manual_seed(6)
    G = torch.randn(dirs.shape[0], 3)
    ref, want = _oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, True)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                      vsize_z=cfg.vsize[2], precision="bf16x3", eval_clamp=False)
    rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    first = rnd.backward(torch.ones_like(G).to(gpu_device), w, N)
    got = rnd.backward(G.to(gpu_device), w, N)
    assert (first["embedding"] - got["embedding"]).abs().max().item() > 0
    for k in ("embedding", "color", "dir"):
        _compare(k, got[k].cpu(), want[k], True)
    for name in MLP_TENSOR_ORDER:
        _compare(name + ".weight", got[name + ".weight"].cpu(), want[name + ".weight"], True)


def test_backward_rejects_bad_arguments(oracle, gpu_device):
    pts = small_scene(20000)
    cfg = oracle_cfg(oracle, SR=16, K=8, P=12)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(8, 8)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=16, K=8, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                      vsize_z=cfg.vsize[2], early_stop_eps=1e-4)
    with pytest.raises(RuntimeError, match="no render call"):
        rnd.backward(torch.zeros(64, 3, device=gpu_device), w, 20000)
    rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    with pytest.raises(RuntimeError, match="early_stop_eps"):
        rnd.backward(torch.zeros(64, 3, device=gpu_device), w, 20000)


def test_backward_with_training_jitter_and_short_rays(oracle, gpu_device):
    """The reference trains at jitter 0.3 (studio_utils.py:166); SR = 8 makes most rays fill every slot, which
    exercises the composite's last-slot segment (`vsize`) and the rays cut at SR in the reverse scan."""
    N, SR, K, P = 80000, 8, 8, 12
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=SR, K=K, P=P)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(28, 28, az=50.0)
    R = dirs.shape[0]
    u = oracle.jitter_uniforms(R, cfg.z_depth_dim, seed=7)
    torch.manual_seed(9)
    G = torch.randn(R, 3)
    ref, want = _oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, True, jitter=0.3, u=u)
    assert (ref["blend_weight"] > 0).sum(-1).max().item() == SR      # some ray uses all SR slots
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                      vsize_z=cfg.vsize[2], precision="fp32", eval_clamp=False, jitter=0.3, seed=7)
    rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    got = rnd.backward(G.to(gpu_device), w, N)
    assert (got["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item() <= 1e-4
    for k in ("embedding", "color", "dir"):
        _compare(k, got[k].cpu(), want[k])
    for name in MLP_TENSOR_ORDER:
        for suf in (".weight", ".bias"):
            _compare(name + suf, got[name + suf].cpu(), want[name + suf])


def test_backward_of_multi_camera_render_is_the_sum_of_the_views(oracle, gpu_device):
    """pnr_render_views + pnr_render_backward with three cameras in one call (shuffled rays, explicit camera
    index) equals the sum of three single-camera backward calls on the same cotangents."""
    N, K = 80000, 8
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, K=K)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, K=K, precision="fp32", eval_clamp=False)
    cams, dirs = [], []
    for az in (15.0, 140.0, 260.0):
        campos, camrot, d = camera_rays(20, 20, az=az)
        cams.append((campos, camrot, 2.0, 6.0))
        dirs.append(d.to(gpu_device))
    n = dirs[0].shape[0]
    torch.manual_seed(2)
    G = torch.randn(3 * n, 3, device=gpu_device)
    total = None
    for v in range(3):
        rnd.render(dirs[v], cams[v][0], cams[v][1], 2.0, 6.0)
        g = rnd.backward(G[v * n:(v + 1) * n], w, N)
        g.pop("rgb")
        total = g if total is None else {k: total[k] + g[k] for k in g}
    perm = torch.randperm(3 * n, generator=torch.Generator().manual_seed(0)).to(gpu_device)
    rnd.render_views(torch.cat(dirs)[perm], cams, n, ray_cam=(perm // n).to(torch.int32))
    got = rnd.backward(G[perm], w, N)
    for k, v in total.items():
        _compare(k, got[k], v)


def test_backward_accumulates_into_caller_buffers(oracle, gpu_device):
    """The C ABI adds to the gradient buffers (torch .grad semantics) and skips null pointers."""
    import ctypes as C
    from pointnerf2studio_amd import _lib
    from pointnerf2studio_amd.renderer import MLP_SHAPES, _ptr, _stream_ptr
    N = 40000
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=24)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=24, precision="fp32", eval_clamp=False)
    campos, camrot, dirs = camera_rays(16, 16, az=35.0)
    rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    G = torch.ones(dirs.shape[0], 3, device=gpu_device)
    once = rnd.backward(G, w, N)
    d, R, arr, n, rc, rpc, cap = rnd._last
    ws_t = [w[nm + ".weight"].to(gpu_device).contiguous() for nm in MLP_TENSOR_ORDER]
    bs_t = [w[nm + ".bias"].to(gpu_device).contiguous() for nm in MLP_TENSOR_ORDER]
    wp = (C.c_void_p * 9)(*[t.data_ptr() for t in ws_t])
    bp = (C.c_void_p * 9)(*[t.data_ptr() for t in bs_t])
    emb = torch.full((N, 32), 1.0, device=gpu_device)
    w0 = torch.full(MLP_SHAPES[0], 2.0, device=gpu_device)
    grads = _lib.GradsC()
    grads.d_embedding = emb.data_ptr()
    grads.d_w[0] = w0.data_ptr()
    lib = _lib.load()
    for _ in range(2):
        _lib.check(lib.pnr_render_backward(scene.handle, wh.handle, C.byref(wp), C.byref(bp), _ptr(d), R, arr, n,
                                           _ptr(rc), rpc, C.byref(rnd.opts), _ptr(G), _ptr(rnd._ws), rnd._ws.numel(),
                                           cap, _ptr(rnd._tws), rnd._tws.numel(), C.byref(grads), None,
                                           _stream_ptr(gpu_device)), "pnr_render_backward")
    torch.cuda.synchronize()
    _compare("embedding (+= twice)", emb - 1.0, 2 * once["embedding"])
    _compare("mlp_base.0.weight (+= twice)", w0 - 2.0, 2 * once["mlp_base.layers.0.weight"])


def test_backward_edge_cases(oracle, gpu_device):
    """Rays that hit nothing (zero rows on the device: every kernel and GEMM split must be a no-op), and K = 20 (the
    generic sample segments of the render, more neighbour slots than a DPP row)."""
    N = 60000
    pts = small_scene(N)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    cfg = oracle_cfg(oracle, SR=24, K=8, P=12)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    for precision in ("fp32", "bf16x3"):
        rnd = RendererHIP(scene, wh, SR=24, K=8, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                          vsize_z=cfg.vsize[2], precision=precision, eval_clamp=False)
        campos, camrot, dirs = camera_rays(16, 16, az=35.0)
        away = -dirs                                             # looking away from the object
        out = rnd.render(away.to(gpu_device), campos, camrot, 2.0, 6.0)
        assert out["counters"]["rays_hit"] == 0
        got = rnd.backward(torch.ones(away.shape[0], 3, device=gpu_device), w, N)
        assert torch.equal(got["rgb"], torch.ones_like(got["rgb"]))
        for k, v in got.items():
            if k != "rgb":
                assert torch.isfinite(v).all() and float(v.abs().max()) == 0.0, k
    # K = 20
    K = 20
    cfg = oracle_cfg(oracle, SR=16, K=K, P=26)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    campos, camrot, dirs = camera_rays(16, 16, az=120.0)
    torch.manual_seed(8)
    G = torch.randn(dirs.shape[0], 3)
    ref, want = _oracle_grads(oracle, pts, w, cfg, campos, camrot, dirs, G, True)
    rnd = RendererHIP(scene, wh, SR=16, K=K, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                      vsize_z=cfg.vsize[2], precision="fp32", eval_clamp=False)
    rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    got = rnd.backward(G.to(gpu_device), w, N)
    assert (got["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item() <= 1e-4
    for k in ("embedding", "color", "dir"):
        _compare(k, got[k].cpu(), want[k])
    for name in MLP_TENSOR_ORDER:
        _compare(name + ".weight", got[name + ".weight"].cpu(), want[name + ".weight"])


def test_point_gradients_are_bitwise_repeatable_and_sparse_rows_match(oracle, gpu_device):
    """The point gradients are a segmented sum in a fixed order (rows grouped by point, ascending row index), not float
    atomics: two backward calls on the same render return the same bits -- also for a crowded cloud in which single
    points are the neighbour of hundreds of samples (groups longer than a wavefront take the selection path).  The sparse
    emission (one row per distinct neighbour point) holds exactly the non-zero rows of the dense tensors."""
    for N, shrink, H in ((60000, 1.0, 24), (300, 0.12, 20)):
        pts = small_scene(N, shrink=shrink)
        cfg = oracle_cfg(oracle, SR=80, K=8, P=12)
        w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
        campos, camrot, dirs = camera_rays(H, H, az=35.0)
        if shrink != 1.0:   # a small object: aim the window at it
            campos, camrot, dirs = camera_rays(800, 800, az=35.0, window=(380, 420, 380, 420))
        scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
        rnd = RendererHIP(scene, wh, SR=80, K=8, D=cfg.z_depth_dim, radius_limit=float(oracle.radius_limit(cfg)),
                          vsize_z=cfg.vsize[2], eval_clamp=False)
        out = rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
        assert out["counters"]["pairs_valid"] > 1000
        G = torch.randn(dirs.shape[0], 3, generator=torch.Generator().manual_seed(8)).to(gpu_device)
        a = rnd.backward(G, w, N)
        b = rnd.backward(G, w, N)
        for k in ("embedding", "color", "dir"):
            assert torch.equal(a[k], b[k]), f"{k} differs between two identical backward calls"
        # ... and so are the MLP gradients (partial tiles + a reducer, no float atomics anywhere in the backward)
        for name in MLP_TENSOR_ORDER:
            for suf in (".weight", ".bias"):
                assert torch.equal(a[name + suf], b[name + suf]), f"{name + suf} differs between two identical calls"
        assert torch.equal(a["rgb"], b["rgb"])
        sp = rnd.backward(G, w, N, sparse_points=True)
        U = out["counters"]["points_unique"]
        assert sp["point_index"].shape == (U,) and sp["point_grads"].shape == (U, 40)
        assert torch.equal(sp["point_index"], rnd.touched_points())
        idx = sp["point_index"]
        assert torch.equal(sp["point_grads"][:, :32], a["embedding"][idx])
        assert torch.equal(sp["point_grads"][:, 32:35], a["color"][idx]) and torch.equal(sp["point_grads"][:, 35:38], a["dir"][idx])
        assert float(sp["point_grads"][:, 38:].abs().sum()) == 0.0
        untouched = torch.ones(N, dtype=torch.bool, device=gpu_device)
        untouched[idx] = False
        assert float(a["embedding"][untouched].abs().sum()) == 0.0
        if shrink != 1.0:
            # the crowded case really has groups beyond one wavefront
            rows = rnd.taps(dirs.shape[0])["smp_pidx"][:out["counters"]["samples_selected"]].reshape(-1)
            assert torch.bincount(rows[rows >= 0]).max().item() > 64, "scene not crowded enough for the long-group path"
