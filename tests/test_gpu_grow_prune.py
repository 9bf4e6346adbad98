"""point growing / pruning and the probing outputs that drive them.

  * pnr_render_probe against the oracle's restatement of models/neural_points_volumetric_model.py:331-352;
  * pnr_scene_update: after a prune (neural_points.py:341-364) and a grow (:367-393) the updated structure answers
    queries exactly as a structure built from nothing on the new cloud, and as the sequential oracle -- with the grid
    unchanged (surviving points reuse their cell code) and with a grown bounding box (everything re-binned);
  * the plugin mirror: PointNerf.prune_points / grow_points / get_probe_outputs, renders after each against the oracle."""
import numpy as np
import pytest
import torch

from helpers import NORTH_STAR, build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
from pointnerf2studio_amd.ns_compat import RayBundle
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, grid_hyperparameters, query_raypos

pytestmark = pytest.mark.gpu


def test_probe_outputs_match_oracle(oracle, gpu_device):
    pts = small_scene(80000)
    cfg = oracle_cfg(oracle, SR=40)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(32, 32, az=75.0)
    ref = oracle.render(pts, w, cfg, campos[None].expand(dirs.shape[0], 3), dirs, 2.0, 6.0, camrot, probe=True)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    rnd = RendererHIP(scene, wh, SR=40)
    out = rnd.render(dirs.to(gpu_device), campos, camrot, 2.0, 6.0)
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
    got = {k: v.cpu() for k, v in rnd.probe().items()}
    want = ref["probe"]
    keep = ref["ray_mask"] > 0
    assert keep.sum().item() > 100
    # rays that are not kept: zeros, index -1
    assert torch.all(got["ray_max_sample_index"][~keep] == -1) and torch.all(got["ray_max_shading_opacity"][~keep] == 0)
    assert torch.all(got["shading_avg_embedding"][~keep] == 0)
    # the arg-max sample: identical unless two opacities of a ray agree to rounding (the GPU's sigma differs from the
    # oracle's in the last bits); such rays are compared on the opacity only
    same = got["ray_max_sample_index"] == want["max_index"]
    assert (same | ~keep).float().mean().item() > 0.98
    assert (got["ray_max_shading_opacity"] - want["max_opacity"]).abs().max().item() <= 1e-5
    sel = keep & same
    names = {"ray_max_sample_loc_w": "max_loc", "ray_max_far_dist": "far_dist", "shading_avg_color": "avg_color",
             "shading_avg_dir": "avg_dir", "shading_avg_conf": "avg_conf", "shading_avg_embedding": "avg_embedding"}
    assert torch.equal(got["ray_max_sample_loc_w"][sel], want["max_loc"][sel])          # positions are exact
    for k, o in names.items():
        err = (got[k][sel] - want[o][sel]).abs().max().item()
        assert err <= 1e-5, f"{k}: {err:.3e}"
    assert want["max_opacity"][keep].max().item() > 0.5 and want["avg_conf"][keep].min().item() > 0


def _query_args(oracle, cfg, pts_xyz, raypos):
    ranges, svs, svd = oracle.get_hyperparameters(cfg, pts_xyz)
    return ranges, svs, svd


@pytest.mark.parametrize("grow_outside", [False, True])
def test_scene_update_equals_fresh_build(oracle, gpu_device, grow_outside):
    pts = small_scene(150000)
    cfg = oracle_cfg(oracle)
    dev = gpu_device
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    raypos, _ = oracle.ray_generation(campos[None], dirs[None], 400, 2.0, 6.0)
    g = torch.Generator().manual_seed(9)

    def hyp_of(xyz):
        return grid_hyperparameters(xyz, cfg.vsize, cfg.vscale, cfg.kernel_size, cfg.ranges)

    def build(xyz):
        h = hyp_of(xyz)
        s = SceneHIP()
        s.build(xyz.to(dev), h.ranges, h.scaled_vsize, h.scaled_vdim, cfg.kernel_size, cfg.query_size, cfg.P, cfg.max_o)
        return s

    scene = build(pts["xyz"])
    # the points that span the bounding box are confident ones here, so that pruning leaves the grid where it is
    # (in general a prune moves the box -- the grid origin is min(xyz) - pad -- and everything is re-binned: the
    # grow_outside = True case below covers that path)
    pts = dict(pts)
    pts["conf"] = pts["conf"].clone()
    for a in range(3):
        pts["conf"][0, pts["xyz"][:, a].argmin(), 0] = 1.0
        pts["conf"][0, pts["xyz"][:, a].argmax(), 0] = 1.0
    # prune by confidence (neural_points.py:341-364), then grow (:367-393)
    pr, old_p = oracle.prune_points(pts, 0.35)
    assert 0.6 * 150000 < pr["xyz"].shape[0] < 0.85 * 150000
    A = 5000
    add_xyz = pr["xyz"][torch.randint(0, pr["xyz"].shape[0], (A,), generator=g)] + (torch.rand(A, 3, generator=g) - 0.5) * 0.004
    if grow_outside:   # a handful of points beyond the old bounding box: the grid origin / dims change
        add_xyz[:8] = torch.tensor([0.6, 0.65, 1.0]) + torch.rand(8, 3, generator=g) * 0.01
    gr, old_g = oracle.grow_points(pr, add_xyz, torch.rand(A, 32, generator=g) - 0.5, torch.rand(A, 3, generator=g),
                                   torch.nn.functional.normalize(torch.randn(A, 3, generator=g), dim=-1),
                                   torch.rand(A, 1, generator=g))
    for step, (cloud, old) in enumerate(((pr, old_p), (gr, old_g))):
        xyz = cloud["xyz"]
        h = hyp_of(xyz)
        info = scene.update(xyz.to(dev), old.to(dev), h.ranges, h.scaled_vsize, h.scaled_vdim, cfg.kernel_size,
                            cfg.query_size, cfg.P, cfg.max_o)
        ui = scene.update_info()
        fresh = build(xyz)
        assert info == {**fresh.info(), "device_bytes": info["device_bytes"]}
        assert ui["updates"] == step + 1 and ui["builds"] == 1
        same_grid = step == 0 or not grow_outside
        if step == 0:
            assert np.array_equal(h.ranges, hyp_of(pts["xyz"]).ranges)
        n_surviving = int((old >= 0).sum())
        assert ui["cells_reused"] == (n_surviving if same_grid else 0), ui
        a = query_raypos(scene, raypos.to(dev), 80, 8, float(oracle.radius_limit(cfg)))
        b = query_raypos(fresh, raypos.to(dev), 80, 8, float(oracle.radius_limit(cfg)))
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x, y)
        ranges, svs, svd = oracle.get_hyperparameters(cfg, xyz)
        rp, rl, rm, st = oracle.query(raypos, xyz[None], cfg.kernel_size, cfg.query_size, 80, 8, svd, cfg.max_o, cfg.P,
                                      oracle.radius_limit(cfg), ranges, svs, True)
        assert torch.equal(a[0].cpu(), rp) and torch.equal(a[1].cpu(), rl) and torch.equal(a[2].cpu(), rm)
        assert st["rays_kept"] > 100


def test_model_prune_grow_probe(oracle, gpu_device):
    pts = small_scene(60000)
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    model = PointNerf(PointNerfConfig(ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=False), point_state_dict=sd).to(gpu_device)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    model.load_state_dict(w, strict=False)
    model.eval()
    model.neural_points.jitter = 0.0
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    R = dirs.shape[0]
    dev = gpu_device
    bundle = RayBundle(origins=campos[None].expand(R, 3).to(dev), directions=dirs.to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev),
                       metadata={"camrotc2w": camrot.reshape(1, 9).expand(R, 9).to(dev)})
    ocfg = oracle_cfg(oracle)

    def check(cloud):
        with torch.no_grad():
            out = model(bundle)
        ref = oracle.render(cloud, w, ocfg, campos[None].expand(R, 3), dirs, 2.0, 6.0, camrot)
        assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
        assert (out["coarse_raycolor"].cpu() - ref["coarse_raycolor"]).abs().max().item() <= NORTH_STAR["rgb"]
        return ref

    check(pts)
    scene = model.neural_points._fused_scene
    # probing outputs under the legacy key names (run/train_studio.py:382)
    probe = model.get_probe_outputs(bundle)
    for k in ("coarse_raycolor", "ray_mask", "ray_max_sample_loc_w", "ray_max_far_dist", "ray_max_shading_opacity",
              "shading_avg_color", "shading_avg_dir", "shading_avg_conf", "shading_avg_embedding"):
        assert k in probe and probe[k].shape[0] == R, k
    # prune (run/train_studio.py:681): the low-confidence points leave, the structure is updated in place
    removed = model.prune_points(0.3)
    pr, _ = oracle.prune_points(pts, 0.3)
    assert removed == 60000 - pr["xyz"].shape[0] and removed > 5000
    assert model.neural_points.points_xyz.shape[0] == pr["xyz"].shape[0]
    assert not model.neural_points.points_xyz.requires_grad and model.neural_points.points_embeding.requires_grad
    check(pr)
    assert model.neural_points._fused_scene is scene and scene.update_info()["updates"] == 1
    # grow (run/train_studio.py:714) from the probe's candidates: high-opacity arg-max samples become points
    keep = (probe["ray_mask"] > 0) & (probe["ray_max_shading_opacity"] > 0.7)
    assert keep.sum().item() > 20
    n = model.grow_points(probe["ray_max_sample_loc_w"][keep], probe["shading_avg_embedding"][keep],
                          probe["shading_avg_color"][keep], probe["shading_avg_dir"][keep],
                          probe["shading_avg_conf"][keep][:, None])
    gr, _ = oracle.grow_points(pr, probe["ray_max_sample_loc_w"][keep].cpu(), probe["shading_avg_embedding"][keep].cpu(),
                               probe["shading_avg_color"][keep].cpu(), probe["shading_avg_dir"][keep].cpu(),
                               probe["shading_avg_conf"][keep][:, None].cpu())
    assert n == gr["xyz"].shape[0] == pr["xyz"].shape[0] + int(keep.sum())
    check(gr)
    assert scene.update_info()["updates"] == 2
    # the grown model trains: gradients reach the appended rows
    model.train()
    out = model(bundle)
    sum(model.get_loss_dict(out, {"image": torch.rand(R, 3, device=dev)}).values()).backward()
    assert model.neural_points.points_embeding.grad.shape[1] == n


def test_prune_after_an_in_place_edit_of_the_cloud_rebuilds_instead_of_updating(oracle, gpu_device):
    """pnr_scene_update reuses the cell codes of surviving points, so it is only valid on a scene built from the cloud
    old_index refers to.  A cloud edited in place since the last render (load_state_dict / copy_ with the same N) with
    an unchanged grid (the config box clamps the ranges) must NOT be updated: the mirror notices the stale key and lets
    the next render build from nothing."""
    pts = small_scene(60000)
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    # a box well inside the cloud: the grid is the (clamped) box whatever the points do
    box = [-0.3, -0.3, -0.3, 0.3, 0.3, 0.3]
    model = PointNerf(PointNerfConfig(ranges=box, max_o=410000, enable_collider=False), point_state_dict=sd).to(gpu_device)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    model.load_state_dict(w, strict=False)
    model.eval()
    model.neural_points.jitter = 0.0
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    R = dirs.shape[0]
    dev = gpu_device
    bundle = RayBundle(origins=campos[None].expand(R, 3).to(dev), directions=dirs.to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev),
                       metadata={"camrotc2w": camrot.reshape(1, 9).expand(R, 9).to(dev)})
    ocfg = oracle_cfg(oracle, ranges=box)
    with torch.no_grad():
        model(bundle)
    scene = model.neural_points._fused_scene
    # the cloud moves in place (same N, same box => same grid), no render in between
    moved = dict(pts)
    moved["xyz"] = (pts["xyz"] * torch.tensor([1.0, -1.0, 1.0]) + torch.tensor([0.01, 0.0, -0.02])).contiguous()
    with torch.no_grad():
        model.neural_points.points_xyz.copy_(moved["xyz"].to(dev))
    removed = model.prune_points(0.3)
    assert removed > 5000
    assert scene.update_info()["updates"] == 0, "a scene built on another cloud was updated"
    pr, _ = oracle.prune_points(moved, 0.3)
    with torch.no_grad():
        out = model(bundle)
    ref = oracle.render(pr, w, ocfg, campos[None].expand(R, 3), dirs, 2.0, 6.0, camrot)
    assert ref["stats"]["rays_kept"] > 20
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"])
    assert (out["coarse_raycolor"].cpu() - ref["coarse_raycolor"]).abs().max().item() <= NORTH_STAR["rgb"]


def test_pack_rows_and_bound_rows_equal_a_full_pack(oracle, gpu_device):
    """pnr_points_pack_rows (a list of rows, its length on the host or on the device) and pnr_points_bind (every render
    refreshes the rows of its own neighbour points from the live tensors) against pnr_points_pack of everything."""
    pts = small_scene(60000)
    cfg = oracle_cfg(oracle, SR=32)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    scene, wh, hyp, info = build_hip(pts, cfg, gpu_device, weights=w)
    dev = gpu_device
    t = {k: pts[k].to(dev).contiguous() for k in ("xyz", "embedding", "conf", "dir", "color")}
    rnd = RendererHIP(scene, wh, SR=32)
    base = rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"].clone()
    touched, count = rnd.touched()
    U = int(count.item())
    assert U == rnd.last_counters["points_unique"] and U > 1000
    idx = touched[:U].long()
    assert torch.equal(idx, rnd.touched_points()) and bool((touched[U:] == touched[0]).all())
    # new features everywhere; the reference image = full pack of the new tensors
    g = torch.Generator().manual_seed(9)
    new = dict(t)
    new["embedding"] = (t["embedding"] + 0.2 * torch.randn(t["embedding"].shape, generator=g).to(dev)).contiguous()
    new["color"] = torch.rand(t["color"].shape, generator=g).to(dev)
    scene.pack_points(new["xyz"], new["embedding"], new["conf"], new["dir"], new["color"])
    want = rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"].clone()
    assert (want - base).abs().max().item() > 1e-2
    # (1) back to the old rows, then ONLY the touched rows re-packed: host-side length, then device-side length
    for kw in ({"index": touched[:U]}, {"index": touched, "count": count}):
        scene.pack_points(t["xyz"], t["embedding"], t["conf"], t["dir"], t["color"])
        scene.pack_point_rows(new["xyz"], new["embedding"], new["conf"], new["dir"], new["color"], **kw)
        got = rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"]
        assert torch.equal(got, want)
    # (2) old rows packed, the NEW tensors bound: the render refreshes what it reads
    scene.pack_points(t["xyz"], t["embedding"], t["conf"], t["dir"], t["color"])
    scene.bind_points(new["xyz"], new["embedding"], new["conf"], new["dir"], new["color"])
    got = rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"].clone()
    assert torch.equal(got, want)
    # ... in place edits of the bound tensors are seen by the next render without any call
    new["color"].mul_(0.5)
    scene.unbind_points()
    scene.pack_points(new["xyz"], new["embedding"], new["conf"], new["dir"], new["color"])
    want2 = rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"].clone()
    new["color"].mul_(2.0)
    scene.bind_points(new["xyz"], new["embedding"], new["conf"], new["dir"], new["color"])
    rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)
    new["color"].mul_(0.5)
    assert torch.equal(rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)["rgb"], want2)
    scene.unbind_points()
    # clear_point_grads: listed rows zeroed, others untouched
    G = torch.ones((60000, 32), device=dev)
    C = torch.ones((60000, 3), device=dev)
    rnd.clear_point_grads(G, C, None, 60000, touched, count)
    assert float(G[idx].abs().sum()) == 0 and float(C[idx].abs().sum()) == 0
    assert int((G.sum(1) == 0).sum()) == U and int((C.sum(1) == 0).sum()) == U
