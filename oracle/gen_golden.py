"""Generates tests/golden/*.npz by RUNNING the reference's own CPU-importable code.

Runs only in the build container (needs /root/reference); the GPU box receives the
committed fixtures, never reference source.  Fixtures are data: seeded inputs and the
reference's outputs.  Recipe: SURVEY.md Appendix B.

  python oracle/gen_golden.py            # rewrites tests/golden/ref_*.npz

What is pinned:
  ref_raygen.npz      near_far_linear_ray_generation   (diff_ray_marching.py:292-336), jitter 0
                      and jitter 0.3 with an injected uniform tensor
  ref_raymarch.npz    ray_march + alpha_blend + radiance_render (diff_ray_marching.py:495-541)
  ref_aggregator.npz  legacy PointAggregator.forward (point_aggregators.py:745-830) configured as
                      the plugin (LeakyReLU slope 0.1, ReLU density, widened sigmoid), seeded weights;
                      also the `weight` and `conf_coefficient` tensors forward returns (:816-830) and a
                      second pass WITH sampled_conf (the legacy weight x clamp(conf) path the probing
                      outputs use)
  ref_trained_aggregator.npz  the same forward with the TRAINED aggregator tensors the reference ships
                      (mvsnet_checkpoints/init/dtu_dgt_d012_img0123_conf_agg2_32_dirclr20/
                      best_net_ray_marching.pth, `aggregator.*` only, stored under the plugin's module
                      names) at LeakyReLU slopes 0.01 (as trained) and 0.1 (the plugin's): sigma of
                      10^2 .. 10^4, the stress SURVEY.md section 7 names
"""
import argparse
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/pointnerf"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_drm():
    spec = importlib.util.spec_from_file_location("ref_drm", f"{REF}/models/rendering/diff_ray_marching.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_aggregator(slope):
    for name in ["nerfstudio", "nerfstudio.utils", "nerfstudio.utils.printing", "nerfstudio.field_components",
                 "nerfstudio.field_components.encodings"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["nerfstudio.utils.printing"].print_tcnn_speed_warning = lambda *a, **k: None
    sys.modules["nerfstudio.field_components.encodings"].NeRFEncoding = object
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if "models" not in sys.modules:
        pkg = types.ModuleType("models")
        pkg.__path__ = [f"{REF}/models"]
        sys.modules["models"] = pkg
    pa = importlib.import_module("models.aggregators.point_aggregators")
    p = argparse.ArgumentParser()
    pa.PointAggregator.modify_commandline_options(p)
    opt = p.parse_args(
        "--agg_dist_pers 20 --agg_distance_kernel linear --agg_intrp_order 2 --act_type LeakyReLU "
        "--shading_feature_mlp_layer1 2 --shading_feature_mlp_layer3 2 --shading_color_mlp_layer 4 "
        "--shading_alpha_mlp_layer 1 --dist_xyz_freq 5 --num_feat_freqs 3 --act_super 0".split())
    for k, v in dict(point_features_dim=32, num_pos_freqs=10, num_viewdir_freqs=4, point_color_mode='1',
                     point_dir_mode='1', sparse_loss_weight=0, zero_one_loss_items=['conf_coefficient'],
                     prob=0).items():
        setattr(opt, k, v)
    agg = pa.PointAggregator(opt)
    for m in agg.modules():
        if isinstance(m, torch.nn.LeakyReLU):
            m.negative_slope = slope
    return agg


# legacy -> plugin module names
NAME_MAP = {
    "block1.0": "mlp_base.layers.0", "block1.2": "mlp_base.layers.1",
    "block3.0": "mlp_head.layers.0", "block3.2": "mlp_head.layers.1",
    "alpha_branch.0": "field_output_density.net",
    "color_branch.0": "mlp_color.layers.0", "color_branch.2": "mlp_color.layers.1",
    "color_branch.4": "mlp_color.layers.2", "color_branch.6": "field_output_color.net",
}


def plugin_weights(agg):
    sd = agg.state_dict()
    return {f"{NAME_MAP[k.rsplit('.', 1)[0]]}.{k.rsplit('.', 1)[1]}": v.clone() for k, v in sd.items()}


def random_aggregator_inputs(seed, R=5, SR=6, K=8):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    a, b = rnd(1).item() * 6.28, rnd(1).item() * 3.14
    ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    Rw2c = torch.tensor([[ca, -sa, 0], [sa * cb, ca * cb, -sb], [sa * sb, ca * sb, cb]], dtype=torch.float32)
    loc_w = rnd(1, R, SR, 3) * 0.5 - 0.25
    s_xyz = loc_w[..., None, :] + (rnd(1, R, SR, K, 3) - 0.5) * 0.03
    # camera-space ("perspective") coordinates: x/z, y/z, z with z around -4 (OpenGL camera looks down -z)
    zc = -3.5 - rnd(1, R, SR, 1)
    loc = torch.cat([(rnd(1, R, SR, 2) - 0.5) * 0.2, zc], dim=-1)
    s_pers = loc[..., None, :] + (rnd(1, R, SR, K, 3) - 0.5) * torch.tensor([0.01, 0.01, 0.03])
    pnt_mask = rnd(1, R, SR, K) > 0.35
    pnt_mask[0, 0, 0, :] = False          # a sample without neighbours
    pnt_mask[0, 1, :, :] = False          # a ray without neighbours
    pnt_mask[0, 2, 1, :] = True
    dirs = torch.nn.functional.normalize(rnd(1, R, 1, 3) - 0.5, dim=-1).expand(-1, -1, SR, -1).contiguous()
    return dict(
        sampled_color=rnd(1, R, SR, K, 3), Rw2c=Rw2c,
        sampled_dir=torch.nn.functional.normalize(rnd(1, R, SR, K, 3) - 0.5, dim=-1),
        sampled_embedding=rnd(1, R, SR, K, 32) - 0.5, sampled_xyz_pers=s_pers, sampled_xyz=s_xyz,
        sample_pnt_mask=pnt_mask, sample_loc=loc, sample_loc_w=loc_w, sample_ray_dirs=dirs)


def run_aggregator(agg, inp, widen=True, conf=None, full=False):
    """forward (point_aggregators.py:745-830).  conf=None is the plugin's call (studio_model.py never multiplies the
    weights by the confidence); with `conf` [1,R,SR,K,1] the legacy weight x clamp(conf) path runs.  full=True also
    returns the `weight` and `conf_coefficient` tensors forward hands back (:816-830)."""
    with torch.no_grad():
        decoded, valid, weight, conf_coefficient = agg(
            inp["sampled_color"], inp["Rw2c"], inp["sampled_dir"], conf, inp["sampled_embedding"].clone(),
            inp["sampled_xyz_pers"], inp["sampled_xyz"], inp["sample_pnt_mask"], inp["sample_loc"],
            inp["sample_loc_w"], inp["sample_ray_dirs"], [0.004] * 3, 0)
        decoded = decoded.clone()
        if widen:   # plugin applies the widened sigmoid unconditionally (studio_model.py:359)
            v = valid[..., None].expand_as(decoded[..., 1:4])
            decoded[..., 1:4] = torch.where(v, decoded[..., 1:4] * (1 + 2 * 0.001) - 0.001, decoded[..., 1:4])
    if full:
        return decoded, valid, weight, conf_coefficient
    return decoded, valid


SHIPPED = f"{REF}/mvsnet_checkpoints/init/dtu_dgt_d012_img0123_conf_agg2_32_dirclr20/best_net_ray_marching.pth"


def trained_fixture():
    """The shipped trained aggregator through the reference's forward on seeded inputs,
    at the slope it was trained with (0.01) and at the plugin's (0.1).  The tensors are stored under the plugin's
    module names (NAME_MAP) so the oracle and the HIP path load them as they load any weights."""
    sd = torch.load(SHIPPED, map_location="cpu", weights_only=True)
    agg_sd = {k[len("aggregator."):]: v.float() for k, v in sd.items() if k.startswith("aggregator.")}
    save = {}
    for slope, tag in ((0.01, "s001"), (0.1, "s01")):
        agg = load_aggregator(slope=slope)
        agg.load_state_dict(agg_sd, strict=True)
        if tag == "s001":
            save.update({f"w_{k}": v for k, v in plugin_weights(agg).items()})
        for case, seed in enumerate([5, 6]):
            inp = random_aggregator_inputs(seed, R=6, SR=8, K=8)
            g = torch.Generator().manual_seed(100 + seed)
            conf = torch.rand(inp["sampled_color"].shape[:-1] + (1,), generator=g) * 1.3 - 0.15   # some outside [1e-4, 1]
            decoded, valid, weight, _ = run_aggregator(agg, inp, full=True)
            decoded_c, _, _, cc = run_aggregator(agg, inp, conf=conf, full=True)
            if tag == "s001":
                save.update({f"c{case}_{k}": v for k, v in inp.items()})
                save[f"c{case}_sampled_conf"] = conf
                save[f"c{case}_valid"] = valid
                save[f"c{case}_weight"] = weight
                save[f"c{case}_conf_coefficient"] = cc
            save[f"c{case}_{tag}_decoded"] = decoded
            save[f"c{case}_{tag}_decoded_conf"] = decoded_c
    np.savez_compressed(os.path.join(OUT, "ref_trained_aggregator.npz"), **to_np(save))


def to_np(d):
    return {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    drm = load_drm()

    # ---- ray generation ------------------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    campos = torch.tensor([[2.0, -3.0, 1.7]])
    raydir = torch.nn.functional.normalize(torch.rand(1, 7, 3, generator=g) - 0.5, dim=-1)
    out = {"campos": campos, "raydir": raydir}
    for D, near, far in [(8, 2.0, 6.0), (400, 2.0, 6.0), (400, 0.1, 8.0)]:
        raypos, seg, valid, tmid = drm.near_far_linear_ray_generation(campos, raydir, D, near=near, far=far, jitter=0.)
        out[f"raypos_D{D}_n{near}_f{far}"] = raypos
        out[f"tmid_D{D}_n{near}_f{far}"] = tmid
    # jitter 0.3 with a known uniform tensor: seed torch's global RNG, record what rand() returns
    torch.manual_seed(7)
    u = torch.rand((1, 7, 400))
    torch.manual_seed(7)
    raypos, seg, valid, tmid = drm.near_far_linear_ray_generation(campos, raydir, 400, near=2.0, far=6.0, jitter=0.3)
    out.update(jit_u=u, jit_raypos=raypos, jit_tmid=tmid)
    np.savez_compressed(os.path.join(OUT, "ref_raygen.npz"), **to_np(out))

    # ---- ray march ------------------------------------------------------------------------
    sys.path.insert(0, REF)   # diff_render_func imports utils.format
    spec = importlib.util.spec_from_file_location("ref_drf", f"{REF}/models/rendering/diff_render_func.py")
    drf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drf)
    g = torch.Generator().manual_seed(3)
    R, SR = 9, 12
    feats = torch.rand(1, R, SR, 4, generator=g)
    feats[..., 0] = feats[..., 0] * torch.tensor([0., 1., 10., 100., 500., 1000., 3000., 50., 5.])[None, :, None]
    ray_valid = torch.rand(1, R, SR, generator=g) > 0.3
    ray_valid[0, 0] = False
    ray_dist = torch.rand(1, R, SR, generator=g) * 0.008 * ray_valid.float()
    (ray_color, point_color, opacity, acc_t, bw, bg_t, _) = drm.ray_march(
        ray_dist, ray_valid, feats, drf.radiance_render, drf.alpha_blend, None)
    np.savez_compressed(os.path.join(OUT, "ref_raymarch.npz"), **to_np(dict(
        ray_dist=ray_dist, ray_valid=ray_valid, feats=feats, ray_color=ray_color, opacity=opacity,
        acc_transmission=acc_t, blend_weight=bw, background_transmission=bg_t)))

    # ---- aggregator (plugin-equivalent: slope 0.1) ------------------------------------------
    torch.manual_seed(1)
    agg = load_aggregator(slope=0.1)
    # weights come from the oracle's seeded generator (so the fixture stores only the seed),
    # loaded INTO the reference module: non-zero biases, density head scaled so sigma is not tiny
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pnr_oracle import make_weights
    w_args = dict(seed=0, sigma_scale=40.0, bias_scale=0.1)
    w = make_weights(**w_args)
    assert set(w) == set(plugin_weights(agg))
    inv = {v: k for k, v in NAME_MAP.items()}
    agg.load_state_dict({f"{inv[k.rsplit('.', 1)[0]]}.{k.rsplit('.', 1)[1]}": v for k, v in w.items()}, strict=True)
    save = {}
    for case, seed in enumerate([1, 2]):
        inp = random_aggregator_inputs(seed)
        decoded, valid, weight, cc = run_aggregator(agg, inp, full=True)
        assert cc == 1        # no sampled_conf: the plugin's call
        save.update({f"c{case}_{k}": v for k, v in inp.items()})
        save[f"c{case}_decoded"] = decoded
        save[f"c{case}_valid"] = valid
        save[f"c{case}_weight"] = weight
        # ... and the legacy weight x clamp(conf) path (what `opt.prob == 1` averages with, :816-830)
        g = torch.Generator().manual_seed(50 + seed)
        conf = torch.rand(inp["sampled_color"].shape[:-1] + (1,), generator=g) * 1.3 - 0.15
        decoded_c, _, weight_c, cc = run_aggregator(agg, inp, conf=conf, full=True)
        assert torch.equal(weight_c, weight)
        save[f"c{case}_sampled_conf"] = conf
        save[f"c{case}_conf_coefficient"] = cc
        save[f"c{case}_decoded_conf"] = decoded_c
    save.update({f"wargs_{k}": np.asarray(v) for k, v in w_args.items()})
    np.savez_compressed(os.path.join(OUT, "ref_aggregator.npz"), **to_np(save))

    # ---- the shipped TRAINED aggregator -------------------------------------------------------
    trained_fixture()

    print("golden fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
