"""SURVEY.md section 8f rank 2 on the GPU path: a checkpoint pair in the layout the reference's trainer writes
(models/base_model.py:85-102: `{epoch}_net_ray_marching.pth` = net.state_dict() with `neural_points.*` and
`aggregator.*` tensors, `{epoch}_states.pth` = {"epoch_count", "total_steps"}) is found and loaded by
PointNerf._init_pointnerf (studio_model.py:147-166), rendered through pnr_render and compared with the CPU oracle."""
import pytest
import torch

from helpers import NORTH_STAR, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
from pointnerf2studio_amd.ns_compat import RayBundle

pytestmark = pytest.mark.gpu


def _write_checkpoint(d, pts, w, epoch, with_aggregator=True):
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    if with_aggregator:
        for src, dst in PointNerf.AGGREGATOR_MAP.items():
            sd[src + ".weight"], sd[src + ".bias"] = w[dst + ".weight"].clone(), w[dst + ".bias"].clone()
    torch.save(sd, d / f"{epoch}_net_ray_marching.pth")
    torch.save({"epoch_count": epoch // 1000, "total_steps": epoch}, d / f"{epoch}_states.pth")


def _bundle(campos, camrot, dirs, device):
    R = dirs.shape[0]
    return RayBundle(origins=campos[None].expand(R, 3).to(device), directions=dirs.to(device),
                     nears=torch.full((R, 1), 2.0, device=device), fars=torch.full((R, 1), 6.0, device=device),
                     metadata={"camrotc2w": camrot.reshape(1, 9).expand(R, 9).to(device)})


def test_checkpoint_pair_loads_and_renders_like_the_oracle(oracle, gpu_device, tmp_path):
    d = tmp_path / "checkpoints" / "chair"
    d.mkdir(parents=True)
    old, new = small_scene(20000, seed=5), small_scene(60000, seed=6)
    w = synthetic.make_weights(2, sigma_scale=300.0, bias_scale=0.1)
    _write_checkpoint(d, old, w, 10000)
    _write_checkpoint(d, new, w, 200000)          # the newest *_states.pth decides (studio_model.py:55-59,156)
    (d / "opt.txt").write_text("not a checkpoint")

    cfg = PointNerfConfig(path_point_cloud=d, ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=False,
                          hip_load_aggregator_weights=True)
    model = PointNerf(cfg).to(gpu_device)
    model.eval()
    model.neural_points.jitter = 0.0
    assert model.neural_points.points_xyz.shape == (60000, 3) and model.neural_points.points_xyz.is_cuda
    assert torch.equal(model.neural_points.points_embeding.detach().cpu(), new["embedding"])
    assert torch.equal(model.mlp_head.layers[1].weight.detach().cpu(), w["mlp_head.layers.1.weight"])

    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    with torch.no_grad():
        out = model(_bundle(campos, camrot, dirs, gpu_device))
    ref = oracle.render(new, w, oracle_cfg(oracle), campos[None].expand(dirs.shape[0], 3), dirs, 2.0, 6.0, camrot)
    assert torch.equal(out["ray_mask"].cpu(), ref["ray_mask"]) and ref["ray_mask"].sum().item() > 50
    assert (out["coarse_raycolor"].cpu() - ref["coarse_raycolor"]).abs().max().item() <= NORTH_STAR["rgb"]
    assert (out["depth"].cpu() - ref["depth"]).abs().max().item() <= NORTH_STAR["depth"]

    # the model's own state_dict round-trips through a file into a second model (nerfstudio's step-*.ckpt path) and
    # renders the same image bit for bit; LPIPS tensors are never part of it (studio_model.py:240-255)
    torch.save(model.state_dict(), tmp_path / "step-000000001.ckpt")
    sd = torch.load(tmp_path / "step-000000001.ckpt", map_location="cpu")
    assert not [k for k in sd if "lpips" in k]
    cfg2 = PointNerfConfig(path_point_cloud=d, ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=False)
    model2 = PointNerf(cfg2).to(gpu_device)        # fresh MLPs (no warm start) ...
    model2.load_state_dict(sd, strict=True)        # ... then the saved weights and features
    model2.eval()
    model2.neural_points.jitter = 0.0
    with torch.no_grad():
        out2 = model2(_bundle(campos, camrot, dirs, gpu_device))
    assert torch.equal(out2["coarse_raycolor"], out["coarse_raycolor"])


def test_checkpoint_without_aggregator_keeps_fresh_mlps(oracle, gpu_device, tmp_path):
    """The reference consumes only neural_points.* (studio_utils.py:84-90); a checkpoint that lacks aggregator.* loads
    as long as the warm start is off, and asking for the warm start then is an error naming the missing tensor."""
    d = tmp_path / "ckpt"
    d.mkdir()
    pts = small_scene(20000, seed=7)
    _write_checkpoint(d, pts, None, 500, with_aggregator=False)
    base = dict(path_point_cloud=d, ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=False)
    model = PointNerf(PointNerfConfig(**base)).to(gpu_device)
    campos, camrot, dirs = camera_rays(16, 16, az=100.0)
    model.eval()
    with torch.no_grad():
        out = model(_bundle(campos, camrot, dirs, gpu_device))
    assert out["coarse_raycolor"].shape == (256, 3) and torch.isfinite(out["coarse_raycolor"]).all()
    with pytest.raises(RuntimeError, match="aggregator.block1.0.weight"):
        PointNerf(PointNerfConfig(hip_load_aggregator_weights=True, **base))
