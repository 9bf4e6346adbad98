"""The plugin surface of the reference (outer surface) reproduced by the host-side mirror:
config fields/defaults, module and parameter names, optimiser groups, callbacks, loss keys.  CPU-only checks of
structure; the GPU behaviour is in test_gpu_plugin.py."""
import dataclasses

import pytest
import torch

from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
from pointnerf2studio_amd.neural_points import PointNeRFEncoding, near_far_linear_ray_generation
from pointnerf2studio_amd.renderer import MLP_SHAPES, MLP_TENSOR_ORDER
from pointnerf2studio_amd import studio_config


def _state_dict(N=500):
    p = synthetic.make_points(N)
    return {"neural_points.xyz": p["xyz"], "neural_points.points_embeding": p["embedding"],
            "neural_points.points_conf": p["conf"], "neural_points.points_dir": p["dir"],
            "neural_points.points_color": p["color"], "neural_points.Rw2c": p["Rw2c"]}


def _cpu_model():
    cfg = PointNerfConfig()
    m = PointNerf.__new__(PointNerf)
    # construct on the CPU for structure checks: same code path with the device overridden
    PointNerf._init_pointnerf_orig = PointNerf._init_pointnerf

    def _init(self):
        self._device = "cpu"
        PointNerf._init_pointnerf_orig(self)
    PointNerf._init_pointnerf = _init
    try:
        m.__init__(cfg, point_state_dict=_state_dict())
    finally:
        PointNerf._init_pointnerf = PointNerf._init_pointnerf_orig
    return m


def test_config_fields_and_defaults_match_reference():
    expect = dict(path_point_cloud=None, eval_num_rays_per_chunk=4096, feat_grad=True, conf_grad=True, dir_grad=True,
                  color_grad=True, num_pos_freqs=10, num_viewdir_freqs=4, num_feat_freqs=3, num_dist_freqs=5,
                  agg_dist_pers=20, point_features_dim=32, point_color_mode=True, point_dir_mode=True, num_samples=80,
                  use_biased_sampler=False, field_dim=64, num_mlp_base_layers=2, num_mlp_head_layers=2,
                  num_color_layers=3, num_alpha_layers=1, hidden_size=256, hidden_size_color=128, apply_pnt_mask=True,
                  act_super=False, axis_weight=[1., 1., 1.], kernel_size=[3, 3, 3], vscale=[2, 2, 2],
                  vsize=[0.004, 0.004, 0.004], query_size=[3, 3, 3], ranges=[-1.2, -1.2, -1.2, 1.2, 1.2, 1.2],
                  z_depth_dim=400, SR=80, K=8, max_o=1000000, P=12, NN=2, gpu_maxthr=1024, zero_epsilon=1e-3,
                  zero_one_loss_weights=0.0001)
    cfg = PointNerfConfig()
    for k, v in expect.items():
        assert getattr(cfg, k) == v, k
    names = {f.name for f in dataclasses.fields(cfg)}
    assert set(expect) <= names


def test_default_arithmetic_is_the_references_fp32():
    """The product computes in the reference's arithmetic unless asked otherwise: fp32 is the default of the plugin
    config, of RendererHIP and of bench.py's headline; bf16x3 (narrower) is opt-in everywhere."""
    import inspect
    import re
    from pathlib import Path
    from pointnerf2studio_amd.renderer import RendererHIP
    assert PointNerfConfig().hip_mlp_mode == "fp32"
    assert inspect.signature(RendererHIP.__init__).parameters["precision"].default == "fp32"
    bench = (Path(__file__).resolve().parent.parent / "bench.py").read_text()
    m = re.search(r'add_argument\("--precision", default="(\w+)"', bench)
    assert m and m.group(1) == "fp32"


def test_missing_point_cloud_raises_like_reference(tmp_path):
    with pytest.raises(RuntimeError, match="does not exist"):
        PointNerfConfig(path_point_cloud=tmp_path / "nope")
    with pytest.raises(RuntimeError, match="must be specified"):
        PointNerf(PointNerfConfig())
    (tmp_path / "empty").mkdir()
    with pytest.raises(RuntimeError, match="Cannot find any _net_ray_marching.pth"):
        PointNerf(PointNerfConfig(path_point_cloud=tmp_path / "empty"))


def test_module_names_shapes_and_param_groups():
    m = _cpu_model()
    sd = m.state_dict()
    for name, shape in zip(MLP_TENSOR_ORDER, MLP_SHAPES):
        assert tuple(sd[name + ".weight"].shape) == shape and tuple(sd[name + ".bias"].shape) == (shape[0],)
    for k, shape in {"neural_points.points_xyz": (500, 3), "neural_points.points_embeding": (1, 500, 32),
                     "neural_points.points_conf": (1, 500, 1), "neural_points.points_dir": (1, 500, 3),
                     "neural_points.points_color": (1, 500, 3), "neural_points.points_Rw2c": (3, 3)}.items():
        assert tuple(sd[k].shape) == shape
    groups = m.get_param_groups()
    assert set(groups) == {"neural_points", "fields"} == set(studio_config.OPTIMIZER_GROUPS)
    named = dict(m.named_parameters())
    np_ids = {id(p) for p in groups["neural_points"]}
    assert np_ids == {id(p) for n, p in named.items() if n.startswith("neural_points.points")}
    assert not m.neural_points.points_xyz.requires_grad and not m.neural_points.points_Rw2c.requires_grad
    assert m.neural_points.points_embeding.requires_grad and m.neural_points.points_color.requires_grad
    assert m.mlp_base.layers[0].in_features == 284 and m.mlp_head.layers[0].in_features == 263
    assert m.mlp_color.layers[0].in_features == 280


def test_training_callbacks_invalidate_packed_copies():
    m = _cpu_model()
    cbs = m.get_training_callbacks(None)
    assert len(cbs) == 1
    m.neural_points._fused_key, m.neural_points._packed_key, m._weights_key = "a", "b", "c"
    cbs[0].run_callback(step=3)
    # the packed copies go stale -- the MFMA-ordered weights for every render, the point rows for the next EVAL render
    # (training renders refresh the rows they read from the bound parameters); the voxel structure stays (points_xyz is
    # frozen) and so does the fact that rows of this cloud exist
    assert m.neural_points._packed_stale and m._weights_key is None
    assert m.neural_points._fused_key == "a" and m.neural_points._packed_key == "b"
    m.neural_points.invalidate()          # a real change of the cloud (grow / prune) drops the structure too
    assert m.neural_points._fused_key is None


def test_unfusable_configuration_raises_instead_of_falling_back_to_torch():
    """The product holds no PyTorch-op render: a network shape the fused kernels do not cover, or hip_fused_training =
    False, raises (checked before any device work).  The op sequence the comparison tests use lives under tests/ and reaches
    the model only through the unfused_outputs_fn hook."""
    from pointnerf2studio_amd.ns_compat import RayBundle
    m = _cpu_model()
    b = RayBundle(origins=torch.zeros(4, 3), directions=torch.zeros(4, 3), nears=torch.full((4, 1), 2.0),
                  fars=torch.full((4, 1), 6.0), metadata={"camrotc2w": torch.eye(3)})
    m.train()
    m.config.hip_fused_training = False
    with pytest.raises(RuntimeError, match="no PyTorch-op render"):
        m.get_outputs(b)
    m.config.hip_fused_training = True
    m.config.num_dist_freqs = 4          # not the shape the kernels are built for
    with pytest.raises(RuntimeError, match="no PyTorch-op render"):
        m.get_outputs(b)
    assert type(m).unfused_outputs_fn is None and not hasattr(PointNerfConfig(), "hip_allow_torch_fallback")
    import pointnerf2studio_amd.model as pm
    assert "cumprod" not in open(pm.__file__).read()       # the composite as torch ops is not in the product
    seen = []
    m.unfused_outputs_fn = lambda model, bundle: seen.append((model, bundle)) or {"hooked": True}
    assert m.get_outputs(b) == {"hooked": True} and seen == [(m, b)]


def test_weighted_conf_loss_equals_the_references_mean():
    """The fused training path returns the reference's conf_coefficient values with multiplicities
    (conf_coefficient_weights) instead of the [1,R'',SR,K] tensor; get_loss_dict's weighted mean is the same number."""
    m = _cpu_model()
    m.train()
    g = torch.Generator().manual_seed(0)
    vals = torch.rand(37, generator=g)
    mult = torch.randint(0, 5, (37,), generator=g).float()
    full = torch.repeat_interleave(vals, mult.long())
    out = {"coarse_raycolor": torch.rand(6, 3), "ray_mask": torch.ones(6, dtype=torch.int8)}
    batch = {"image": torch.rand(6, 3)}
    a = m.get_loss_dict({**out, "conf_coefficient": full}, batch)["conf_coefficient_loss"]
    b = m.get_loss_dict({**out, "conf_coefficient": vals, "conf_coefficient_weights": mult}, batch)["conf_coefficient_loss"]
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)


def test_shim_chunk_loop_and_collider():
    """ns_compat (no nerfstudio in the image): get_outputs_for_camera_ray_bundle cuts the [H, W] bundle into row-major
    chunks of eval_num_rays_per_chunk rays through forward() and views the concatenation as [H, W, -1]; forward applies
    the collider."""
    from pointnerf2studio_amd import ns_compat as ns
    if ns.HAVE_NERFSTUDIO:
        pytest.skip("the real nerfstudio classes are in use")

    class Probe(ns.Model):
        def get_outputs(self, rb):
            self.seen.append((len(rb), float(rb.nears.flatten()[0]), float(rb.fars.flatten()[0])))
            return {"x": rb.origins[..., :1] * 2.0, "n": 3}

    cfg = ns.ModelConfig(eval_num_rays_per_chunk=7)
    mod = Probe(cfg)
    mod.seen = []
    H, W = 4, 5
    o = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
    cam = ns.RayBundle(origins=o, directions=torch.zeros(H, W, 3), metadata={"camrotc2w": torch.zeros(H, W, 9)})
    mod.eval()
    out = mod.get_outputs_for_camera_ray_bundle(cam)
    assert [s[0] for s in mod.seen] == [7, 7, 6] and set(out) == {"x"}
    assert out["x"].shape == (H, W, 1) and torch.equal(out["x"], o[..., :1] * 2.0)
    assert all(s[1] == 0.0 and s[2] == 6.0 for s in mod.seen)       # eval: near plane reset to 0 [ns-mem]
    mod.train()
    mod.seen = []
    mod(cam.get_row_major_sliced_ray_bundle(0, 3))
    assert mod.seen == [(3, 2.0, 6.0)]


def test_loss_dict_keys_and_values():
    m = _cpu_model()
    m.train()
    out = {"coarse_raycolor": torch.rand(10, 3), "ray_mask": torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 1, 0], dtype=torch.int8),
           "conf_coefficient": torch.rand(1, 4, 5, 8)}
    batch = {"image": torch.rand(10, 3)}
    ld = m.get_loss_dict(out, batch)
    assert set(ld) == {"ray_masked_coarse_raycolor_loss", "conf_coefficient_loss"}
    keep = out["ray_mask"] > 0
    assert torch.allclose(ld["ray_masked_coarse_raycolor_loss"],
                          torch.nn.functional.mse_loss(batch["image"][keep], out["coarse_raycolor"][keep]) + 1e-6)
    m.eval()
    assert set(m.get_loss_dict(out, batch)) == {"ray_masked_coarse_raycolor_loss"}


def test_encoding_and_ray_generation_mirrors_match_oracle(oracle):
    x = torch.rand(7, 5) - 0.5
    for F, ori in [(3, False), (5, False), (4, True)]:
        assert torch.equal(PointNeRFEncoding(5, F, ori)(x), oracle.positional_encoding(x, F, ori))
    campos = torch.tensor([[1.0, 2.0, 3.0]])
    raydir = torch.nn.functional.normalize(torch.rand(1, 6, 3) - 0.5, dim=-1)
    rp, _, _, tm = near_far_linear_ray_generation(campos, raydir, 400, near=2.0, far=6.0, jitter=0.0)
    rp_o, tm_o = oracle.ray_generation(campos, raydir, 400, 2.0, 6.0)
    assert torch.equal(rp, rp_o) and torch.equal(tm, tm_o)


def test_method_registration_constants():
    assert studio_config.METHOD_NAME == "pointnerf-original"
    assert studio_config.OPTIMIZER_GROUPS == {"fields": 0.0005, "neural_points": 0.002}
    assert studio_config.EVAL_NUM_RAYS_PER_CHUNK == 2304
    f = studio_config.pointnerf_lr_lambda()
    assert f(0) == 1.0 and abs(f(1000000) - 0.1) < 1e-12


def test_loads_reference_checkpoint_layout(tmp_path, monkeypatch):
    """studio_model.py:147-166 / base_model.py:85-102: `{iter}_net_ray_marching.pth` + `{iter}_states.pth`; the
    newest iteration wins, only the neural_points.* keys are consumed, aggregator.* weights are ignored."""
    sd_old, sd_new = _state_dict(300), _state_dict(500)
    sd_new["aggregator.block1.0.weight"] = torch.zeros(256, 284)      # present in real checkpoints, unused
    d = tmp_path / "ckpt"
    d.mkdir()
    torch.save(sd_old, d / "0_net_ray_marching.pth")
    torch.save({"epoch_count": 0, "total_steps": 0}, d / "0_states.pth")
    torch.save(sd_new, d / "200000_net_ray_marching.pth")
    torch.save({"epoch_count": 5, "total_steps": 200000}, d / "200000_states.pth")
    orig = PointNerf._init_pointnerf

    def _init(self):
        self._device = "cpu"
        orig(self)
    monkeypatch.setattr(PointNerf, "_init_pointnerf", _init)
    m = PointNerf(PointNerfConfig(path_point_cloud=d))
    assert m.neural_points.points_xyz.shape == (500, 3)
    assert torch.equal(m.neural_points.points_embeding.detach(), sd_new["neural_points.points_embeding"])
    assert not any(k.startswith("aggregator") for k in m.state_dict())


def test_optional_aggregator_warm_start(tmp_path, monkeypatch):
    """hip_load_aggregator_weights (opt-in): the legacy checkpoint's `aggregator.*` Linear
    layers initialise the plugin MLPs by the name map of SURVEY.md section 8c; a missing or mis-shaped tensor is an
    error and leaves the model untouched."""
    from pointnerf2studio_amd import synthetic
    sd = _state_dict(400)
    w = synthetic.make_weights(3, sigma_scale=1.0, bias_scale=0.2)
    for src, dst in PointNerf.AGGREGATOR_MAP.items():
        sd[src + ".weight"] = w[dst + ".weight"].clone()
        sd[src + ".bias"] = w[dst + ".bias"].clone()
    d = tmp_path / "ckpt"
    d.mkdir()
    torch.save(sd, d / "100_net_ray_marching.pth")
    torch.save({"epoch_count": 1, "total_steps": 100}, d / "100_states.pth")
    orig = PointNerf._init_pointnerf

    def _init(self):
        self._device = "cpu"
        orig(self)
    monkeypatch.setattr(PointNerf, "_init_pointnerf", _init)
    plain = PointNerf(PointNerfConfig(path_point_cloud=d))
    assert not torch.equal(plain.mlp_base.layers[0].weight.detach(), w["mlp_base.layers.0.weight"])
    m = PointNerf(PointNerfConfig(path_point_cloud=d, hip_load_aggregator_weights=True))
    for name in w:
        assert torch.equal(m.state_dict()[name], w[name]), name
    # a checkpoint without the colour head: refused, nothing half-loaded
    bad = dict(sd)
    del bad["aggregator.color_branch.6.weight"]
    before = {k: v.clone() for k, v in plain.state_dict().items()}
    with pytest.raises(RuntimeError, match="color_branch.6.weight"):
        plain.load_aggregator_weights(bad)
    bad = dict(sd)
    bad["aggregator.block3.0.weight"] = torch.zeros(256, 256)
    with pytest.raises(RuntimeError, match="shape"):
        plain.load_aggregator_weights(bad)
    assert all(torch.equal(v, plain.state_dict()[k]) for k, v in before.items())


def test_aggregator_map_against_the_shipped_checkpoints():
    """tests/golden/shipped_checkpoint_keys.json lists key -> shape of the reference's own
    mvsnet_checkpoints/init/*/best_net_ray_marching.pth (generated in the build container by
    oracle/gen_checkpoint_listing.py).  The warm-start name map must name tensors that exist there with the plugin's
    shapes for the 32-feature `agg2_32_dirclr20` variant (the one the reference's scripts train from); the other
    shipped variant (501-input first layer: 63-dim positional features) must be refused by shape."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shipped_checkpoint_keys.json")) as f:
        listing = json.load(f)["checkpoints"]
    good = listing["dtu_dgt_d012_img0123_conf_agg2_32_dirclr20"]
    by_dst = dict(zip(MLP_TENSOR_ORDER, MLP_SHAPES))
    for src, dst in PointNerf.AGGREGATOR_MAP.items():
        assert good[src + ".weight"]["shape"] == list(by_dst[dst]), (src, dst)
        assert good[src + ".bias"]["shape"] == [by_dst[dst][0]]
        assert good[src + ".weight"]["dtype"] == "float32"
    assert set(PointNerf.AGGREGATOR_MAP.values()) == set(MLP_TENSOR_ORDER)
    assert {k.rsplit(".", 1)[0] for k in good if k.startswith("aggregator.")} == set(PointNerf.AGGREGATOR_MAP)
    # the shipped initialisation checkpoints carry no point cloud (it comes from the per-scene training checkpoint)
    assert not [k for k in good if k.startswith("neural_points.")]
    other = listing["dtu_dgt_d012_img0123_conf_color_dir_agg2"]
    assert other["aggregator.block1.0.weight"]["shape"] == [256, 501]
    m = _cpu_model()
    sd = {k: torch.zeros(v["shape"]) for k, v in other.items()}
    with pytest.raises(RuntimeError, match="shape"):
        m.load_aggregator_weights(sd)
    # ... while the listed good variant is accepted tensor for tensor
    assert m.load_aggregator_weights({k: torch.full(v["shape"], 0.5) for k, v in good.items()}) == 18
    assert float(m.mlp_head.layers[0].weight.detach().mean()) == 0.5
