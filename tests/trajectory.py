"""Shared pieces of the N-step training-trajectory tests (tests/test_gpu_training_trajectory.py, test infrastructure):
the SAME training run -- a student network + point features fitted to images a teacher rendered -- driven once through
autograd over the CPU oracle and once through the plugin mirror on the HIP path.

What `ns-train pointnerf-original` does per step (studio_model.py:263-431, studio_config.py:33-48, the datamanager's one
image per batch, studio_datamanager.py:62-81): forward with the 0.3 coarse-sample jitter, get_loss_dict, backward, Adam
on two parameter groups ("fields" lr 5e-4, "neural_points" lr 2e-3; nerfstudio's AdamOptimizerConfig: eps 1e-8, no
weight decay [ns-mem]) with the exponential decay of studio_utils.py:33-44, the after-step callbacks.  Both sides draw the
same jitter uniforms: the HIP path's counter-based generator keyed on (seed = index of the model's render call, ray,
sample) and its numpy restatement `pnr_oracle.jitter_uniforms` fed to the oracle's ray generation.
"""
import math

import torch

from helpers import camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic

LR = {"fields": 5e-4, "neural_points": 2e-3}                  # studio_config.py:33-48
POINT_KEYS = ("embedding", "conf", "dir", "color")           # the trainable point tensors (studio_utils.py:84-90)
JITTER = 0.3                                                  # studio_utils.py:166


def lr_lambda(step):                                          # PointNerfScheduler, studio_utils.py:33-44
    return pow(0.1, step / 1000000)


def make_problem(oracle, N=40000, H=32, W=32, azimuths=(35.0, 125.0), SR=32, K=8, student_seed=5):
    """Teacher weights (seed 0) render the target images of `azimuths` through the oracle (eval composite, jitter 0); the
    student starts from other weights AND other point features (colour / embedding perturbed)."""
    pts = small_scene(N)
    cfg = oracle_cfg(oracle, SR=SR, K=K)
    teacher = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    views = []
    for az in azimuths:
        campos, camrot, dirs = camera_rays(H, W, az=az)
        ref = oracle.render(pts, teacher, cfg, campos[None].expand(dirs.shape[0], 3), dirs, 2.0, 6.0, camrot)
        views.append({"campos": campos, "camrot": camrot, "dirs": dirs, "target": ref["coarse_raycolor"].clone()})
    student = synthetic.make_weights(student_seed, sigma_scale=300.0, bias_scale=0.1)
    g = torch.Generator().manual_seed(student_seed)
    spts = {k: v.clone() for k, v in pts.items()}
    spts["color"] = torch.rand(pts["color"].shape, generator=g)
    spts["embedding"] = pts["embedding"] + 0.1 * (torch.rand(pts["embedding"].shape, generator=g) - 0.5)
    return {"cfg": cfg, "views": views, "points": spts, "weights": student, "SR": SR, "K": K, "H": H, "W": W}


def growth(points, n_add=1500, seed=9):
    """Seeded stand-in for what probe_hole hands to grow_points (run/train_studio.py:676-735): new points next to existing
    ones.  Returns (add_xyz, add_embedding, add_color, add_dir, add_conf)."""
    g = torch.Generator().manual_seed(seed)
    N = points["xyz"].shape[0]
    pick = torch.randperm(N, generator=g)[:n_add]
    xyz = points["xyz"][pick] + 0.003 * (torch.rand((n_add, 3), generator=g) - 0.5)
    emb = torch.rand((n_add, 32), generator=g) - 0.5
    color = torch.rand((n_add, 3), generator=g)
    d = torch.nn.functional.normalize(torch.rand((n_add, 3), generator=g) - 0.5, dim=-1)
    conf = 0.5 + 0.5 * torch.rand((n_add, 1), generator=g)
    return xyz.contiguous(), emb, color, d, conf


def psnr(a, b):
    return float(-10.0 * math.log10(float(((a - b) ** 2).mean()) + 1e-20))


def _oracle_optimizer(points, weights):
    opt = torch.optim.Adam([{"params": list(weights.values()), "lr": LR["fields"]},
                            {"params": [points[k] for k in POINT_KEYS], "lr": LR["neural_points"]}], eps=1e-8)
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)


def eval_images(oracle, prob, points, weights):
    out = []
    with torch.no_grad():
        for v in prob["views"]:
            R = v["dirs"].shape[0]
            out.append(oracle.render(points, weights, prob["cfg"], v["campos"][None].expand(R, 3), v["dirs"], 2.0, 6.0,
                                     v["camrot"])["coarse_raycolor"])
    return out


def run_oracle(oracle, prob, steps, seeds, edit_at=None, prune_thresh=0.25):
    """`steps` optimiser steps through torch autograd over the CPU oracle.  seeds[i]: the jitter seed of step i (the HIP
    model's render-call index).  edit_at: prune + grow between steps edit_at - 1 and edit_at, optimisers re-created (as
    the reference's trainer does, run/train_studio.py:676-684,714-716).  Returns (losses, points, weights)."""
    cfg = prob["cfg"]
    points = {k: v.clone() for k, v in prob["points"].items()}
    weights = {k: v.clone().requires_grad_(True) for k, v in prob["weights"].items()}
    for k in POINT_KEYS:
        points[k].requires_grad_(True)
    opt, sched = _oracle_optimizer(points, weights)
    losses = []
    # (a GPU box gives a test 16 CPUs of a 256-thread host: torch's default pool of one thread per logical CPU is what
    # makes these small GEMMs slow there)
    threads_before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, threads_before)))
    try:
        return _run_oracle_steps(oracle, prob, cfg, points, weights, opt, sched, losses, steps, seeds, edit_at, prune_thresh)
    finally:
        torch.set_num_threads(threads_before)


def _run_oracle_steps(oracle, prob, cfg, points, weights, opt, sched, losses, steps, seeds, edit_at, prune_thresh):
    for i in range(steps):
        if edit_at is not None and i == edit_at:
            with torch.no_grad():
                det = {k: v.detach() for k, v in points.items()}
                det, _ = oracle.prune_points(det, prune_thresh)
                det, _ = oracle.grow_points(det, *growth(det))
            points = {k: v.clone() for k, v in det.items()}
            for k in POINT_KEYS:
                points[k].requires_grad_(True)
            opt, sched = _oracle_optimizer(points, weights)
        v = prob["views"][i % len(prob["views"])]
        R = v["dirs"].shape[0]
        u = oracle.jitter_uniforms(R, cfg.z_depth_dim, seeds[i])
        opt.zero_grad(set_to_none=True)
        out = oracle.render(points, weights, cfg, v["campos"][None].expand(R, 3), v["dirs"], 2.0, 6.0, v["camrot"],
                            jitter=JITTER, u=u, training=True)
        loss = sum(oracle.get_loss_dict(out, v["target"], training=True).values())
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    return losses, {k: v.detach() for k, v in points.items()}, {k: v.detach() for k, v in weights.items()}


# ---- the HIP side: the plugin mirror as `ns-train pointnerf-original` configures it ------------------------------------
def make_model(prob, device):
    from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
    p = prob["points"]
    sd = {"neural_points.xyz": p["xyz"], "neural_points.points_embeding": p["embedding"],
          "neural_points.points_conf": p["conf"], "neural_points.points_dir": p["dir"],
          "neural_points.points_color": p["color"], "neural_points.Rw2c": p["Rw2c"]}
    # (studio_config.py: the datamanager's planes through nerfstudio's collider, one camera per bundle)
    cfg = PointNerfConfig(ranges=list(synthetic.CHAIR_RANGES), max_o=prob["cfg"].max_o, SR=prob["SR"], K=prob["K"],
                          enable_collider=True, collider_params={"near_plane": 2.0, "far_plane": 6.0},
                          hip_single_camera_bundles=True)
    model = PointNerf(cfg, point_state_dict=sd).to(device)
    missing = model.load_state_dict(prob["weights"], strict=False)
    assert not missing.unexpected_keys
    if hasattr(model.collider, "reset_near_plane"):
        model.collider.reset_near_plane = False       # eval renders keep the datamanager's near plane
    return model


def _hip_optimizer(model):
    groups = model.get_param_groups()
    opt = torch.optim.Adam([{"params": groups[name], "lr": lr} for name, lr in LR.items()], eps=1e-8)
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)


def _bundle(v, device):
    from pointnerf2studio_amd.ns_compat import RayBundle
    R = v["dirs"].shape[0]
    return RayBundle(origins=v["campos"][None].expand(R, 3).to(device), directions=v["dirs"].to(device),
                     metadata={"camrotc2w": v["camrot"].to(device)})


def hip_state(model):
    npts = model.neural_points
    pts = {"xyz": npts.points_xyz, "embedding": npts.points_embeding, "conf": npts.points_conf, "dir": npts.points_dir,
           "color": npts.points_color, "Rw2c": npts.points_Rw2c}
    w = {k: v for k, v in model.state_dict().items() if k.split(".")[0] in
         ("mlp_base", "mlp_head", "mlp_color", "field_output_density", "field_output_color")}
    return {k: v.detach().cpu().clone() for k, v in pts.items()}, {k: v.detach().cpu().clone() for k, v in w.items()}


def hip_eval_images(model, prob, device):
    was, jit = model.training, model.neural_points.jitter
    model.eval()
    model.neural_points.jitter = 0.0
    out = []
    with torch.no_grad():
        for v in prob["views"]:
            out.append(model(_bundle(v, device))["coarse_raycolor"].cpu().clone())
    model.neural_points.jitter = jit
    model.train(was)
    return out


def run_hip(prob, steps, device, edit_at=None, prune_thresh=0.25, model=None, hook=None):
    """The same run through PointNerf.forward + get_loss_dict + backward + torch.optim.Adam + the after-step callbacks on
    the fused HIP path.  Returns (losses, jitter seeds used, model).  hook(i, model): called before step i (tests)."""
    model = make_model(prob, device) if model is None else model
    model.train()
    assert model.neural_points.jitter == JITTER
    opt, sched = _hip_optimizer(model)
    callbacks = model.get_training_callbacks(None)
    targets = [v["target"].to(device) for v in prob["views"]]
    losses, seeds = [], []
    for i in range(steps):
        if edit_at is not None and i == edit_at:
            model.prune_points(prune_thresh)
            host = {"xyz": model.neural_points.points_xyz.detach().cpu()}
            model.grow_points(*growth(host))
            opt, sched = _hip_optimizer(model)          # the parameters are new tensors
        if hook is not None:
            hook(i, model)
        k = i % len(prob["views"])
        seeds.append(model._render_calls & 0xFFFFFFFF)
        opt.zero_grad(set_to_none=True)
        out = model(_bundle(prob["views"][k], device))          # a NEW bundle object per step, as the datamanager's
        loss = sum(model.get_loss_dict(out, {"image": targets[k]}).values())
        loss.backward()
        opt.step()
        sched.step()
        for cb in callbacks:
            cb.run_callback(step=i)
        losses.append(loss.detach())
    return [float(x) for x in torch.stack(losses).cpu()], seeds, model


def perturbed(prob, eps=1e-7, seed=3):
    """The same problem with every embedding value moved by eps relative (half a float32 ulp): what a different but equally
    valid float32 evaluation order amounts to.  Used to measure how far two such runs drift apart BY THEMSELVES."""
    g = torch.Generator().manual_seed(seed)
    pts = {k: v.clone() for k, v in prob["points"].items()}
    pts["embedding"] = pts["embedding"] * (1 + eps * (torch.rand(pts["embedding"].shape, generator=g) - 0.5))
    out = dict(prob)
    out["points"] = pts
    return out
