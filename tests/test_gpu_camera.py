"""rays from cameras inside the kernels (pnr_render_camera).  The reference's datamanager
hands the model ray bundles nerfstudio's generator materialised (studio_datamanager.py:62-110: origins [R,3],
directions [R,3], and a per-ray copy of the rotation at :108); here a view is pose + intrinsics and a ray is a pixel id.

  * the directions the kernels generate (pnr_camera_rays writes them out) equal the host statement pnr_pinhole_ray
    bit for bit, for every pixel of a 1600 x 1200 frame with off-centre principal point and fx != fy;
  * pnr_render_camera equals pnr_render_views fed with those directions bit for bit (image, depth, mask, counters,
    neighbour lists), for the whole frame, for a tile shard's pixel list and for several views in one call;
  * against the CPU oracle on the same rays: index lists exact, image within the north_star bar;
  * a training step (pnr_render_backward) after a camera render equals the one after a render from the tensor."""
import numpy as np
import pytest
import torch

from helpers import NORTH_STAR, build_hip, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.distributed import make_shard
from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER, RendererHIP, View, camera_rays, pinhole_ray

pytestmark = pytest.mark.gpu


def _views(H, W, azimuths, angle_x=0.6911112070083618):
    out = []
    for az in azimuths:
        campos, camrot = synthetic.make_camera(az)
        out.append(View.from_angle(campos, camrot, H, W, angle_x))
    return out


def test_device_rays_equal_the_host_statement(gpu_device):
    H, W = 1200, 1600
    campos, camrot = synthetic.make_camera(70.0, 15.0)
    v = View(campos, camrot, fx=1650.0, fy=1660.5, cx=790.25, cy=611.5)
    d = camera_rays([v], H, W, gpu_device).cpu().numpy().reshape(H, W, 3)
    rng = np.random.RandomState(0)
    ys, xs = rng.randint(0, H, 4000), rng.randint(0, W, 4000)
    for y, x in list(zip(ys, xs)) + [(0, 0), (H - 1, W - 1), (0, W - 1), (H - 1, 0)]:
        assert np.array_equal(d[y, x], pinhole_ray(v, int(x), int(y))), (x, y)
    # unit length, and the nerfstudio convention (synthetic.make_rays states it with torch ops) to rounding
    assert np.abs(np.linalg.norm(d.astype(np.float64), axis=-1) - 1).max() < 1e-6
    v2 = View.from_angle(campos, camrot, H, W, 0.9)
    d2 = camera_rays([v2], H, W, gpu_device).cpu()
    assert (d2 - synthetic.make_rays(H, W, campos, camrot, 0.9)).abs().max().item() < 3e-7
    # a pixel list: rows follow the list, view-major
    px = torch.tensor([5, 0, W * H - 1, W + 3], dtype=torch.int32, device=gpu_device)
    d3 = camera_rays([v2, v], H, W, gpu_device, pixels=px).cpu().numpy()
    assert np.array_equal(d3[:4], d2.numpy()[[5, 0, W * H - 1, W + 3]])
    assert np.array_equal(d3[4:], d.reshape(-1, 3)[[5, 0, W * H - 1, W + 3]])


@pytest.mark.parametrize("precision,jitter", [("fp32", 0.0), ("fp32", 0.3), ("bf16x3", 0.0)])
def test_render_camera_equals_render_from_the_direction_tensor(oracle, gpu_device, precision, jitter):
    pts = small_scene(120000)
    cfg = oracle_cfg(oracle)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    H, W = 48, 64
    views = _views(H, W, [35.0, 160.0, 290.0])

    def both(pixels):
        rnd = RendererHIP(scene, wh, precision=precision, jitter=jitter, seed=5)
        a = rnd.render_camera(views, H, W, pixels=pixels)
        a = {k: (v.clone() if torch.is_tensor(v) else dict(v)) for k, v in a.items()}
        S = a["counters"]["samples_selected"]
        lists_a = rnd.taps(a["rgb"].shape[0])["smp_pidx"][:S].clone()
        dirs = camera_rays(views, H, W, gpu_device, pixels=pixels)
        n = dirs.shape[0] // len(views)
        if jitter > 0 and pixels is not None:
            # the jitter stream of a ray from a camera is keyed on (view, pixel id), of a ray from a tensor on its index
            # in the call: the two coincide for whole frames only.  What a pixel list must equal under jitter is the
            # whole-frame render at those pixels (below).
            return a, dirs
        b = rnd.render_views(dirs, [(v.campos, v.camrotc2w, v.near, v.far) for v in views], n)
        lists_b = rnd.taps(b["rgb"].shape[0])["smp_pidx"][:S]
        for k in ("rgb", "depth", "acc", "ray_mask"):
            assert torch.equal(a[k], b[k]), k
        assert a["counters"] == b["counters"] and torch.equal(lists_a, lists_b)
        return a, dirs

    full, dirs_full = both(None)
    assert full["counters"]["rays_kept"] > 300
    # a rank's tile shard (16 x 16 tiles dealt round-robin to 3 ranks): the same pixels, in shard order
    shard = make_shard(H, W, 3, 1)
    px = shard.pixels.to(torch.int32).to(gpu_device)
    part, _ = both(px)
    n_px = px.numel()
    for v in range(len(views)):
        # (also under jitter: a pixel draws the same uniforms in every cut of the frame)
        assert torch.equal(part["rgb"][v * n_px:(v + 1) * n_px],
                           full["rgb"][v * H * W:(v + 1) * H * W][shard.pixels.to(gpu_device)])
    # one pixel list PER VIEW (pnr_render_camera_lists, the rotated shard of a multi-GPU step): view i renders the tiles
    # of owner (1 + i) % 3 -- every view's rows equal the whole-frame render at those pixels, with or without jitter
    rsh = make_shard(H, W, 3, 1, rotate=True)
    rnd = RendererHIP(scene, wh, precision=precision, jitter=jitter, seed=5)
    lists = rsh.view_pixels[:len(views)].to(torch.int32).to(gpu_device)
    per_view = rnd.render_camera(views, H, W, pixels=lists)
    assert per_view["rgb"].shape[0] == len(views) * rsh.n_pad
    for v in range(len(views)):
        rows = slice(v * rsh.n_pad, (v + 1) * rsh.n_pad)
        at = rsh.pixels_of_view(v).to(gpu_device)
        assert torch.equal(per_view["rgb"][rows], full["rgb"][v * H * W:(v + 1) * H * W][at])
        assert torch.equal(per_view["depth"][rows], full["depth"][v * H * W:(v + 1) * H * W][at])
        assert torch.equal(per_view["ray_mask"][rows], full["ray_mask"][v * H * W:(v + 1) * H * W][at])
    assert not torch.equal(rsh.view_pixels[0], rsh.view_pixels[1])
    if jitter == 0.0 and precision == "fp32":
        # ... and the oracle on the same rays (view 1)
        v = views[1]
        d1 = dirs_full[H * W:2 * H * W].cpu()
        ref = oracle.render(pts, w, cfg, torch.as_tensor(v.campos)[None].expand(H * W, 3), d1, v.near, v.far,
                            torch.as_tensor(v.camrotc2w))
        sl = slice(H * W, 2 * H * W)
        assert torch.equal(full["ray_mask"][sl].cpu(), ref["ray_mask"])
        assert (full["rgb"][sl].cpu() - ref["coarse_raycolor"]).abs().max().item() <= NORTH_STAR["rgb"]
        assert (full["depth"][sl].cpu() - ref["depth"]).abs().max().item() <= NORTH_STAR["depth"]


def test_backward_after_camera_render(oracle, gpu_device):
    """The training step after pnr_render_camera differentiates the same render: pnr_render_backward takes the
    directions k_expand left in the workspace (pnr_render_taps.ray_dirs) and returns the gradients it returns after a
    render from the direction tensor."""
    pts = small_scene(60000)
    cfg = oracle_cfg(oracle)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    H, W = 24, 24
    views = _views(H, W, [35.0, 200.0])
    G = torch.randn(2 * H * W, 3, generator=torch.Generator().manual_seed(4)).to(gpu_device)
    N = pts["xyz"].shape[0]
    rnd = RendererHIP(scene, wh, eval_clamp=False)
    rnd.render_camera(views, H, W)
    got = rnd.backward(G, w, N)
    dirs = camera_rays(views, H, W, gpu_device)
    rnd.render_views(dirs, [(v.campos, v.camrotc2w, v.near, v.far) for v in views], H * W)
    want = rnd.backward(G, w, N)
    assert got["embedding"].abs().sum().item() > 0
    for k in ["embedding", "color", "dir", "rgb"] + [n + ".weight" for n in MLP_TENSOR_ORDER]:
        scale = want[k].abs().max().item()
        # the same arithmetic on the same numbers; float atomics in the point scatter may reorder sums
        assert (got[k] - want[k]).abs().max().item() <= 1e-5 * scale + 1e-12, k
