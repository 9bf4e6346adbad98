"""GPU box: a small `ns-train pointnerf-original` in one file -- the plugin mirror trained for a few hundred steps the way
nerfstudio's trainer drives it (studio_config.py:17-48, studio_datamanager.py:62-110): every step ONE image of the training
set, 4096 random pixels of it, forward with the 0.3 jitter, get_loss_dict, backward, the registered optimisers (torch Adam
5e-4 for the MLPs, PointRowAdam 2e-3 for the point tensors) with the exponential schedule, the after-step callbacks; every
`--eval-every` steps the eval images of ALL views through get_outputs_for_camera_ray_bundle (one fused call each).

    python tools/train_demo.py [--points 500000] [--steps 600] [--side 200] [--views 8]

Data is synthetic (no dataset is reachable): a TEACHER network + point features render the training images through the same
HIP path; the student starts from other weights, random colours and perturbed embeddings.  What the run shows is the
training LOOP at scale -- many transitions between training renders (bound point rows, taped renders, dense gradient
buffers cleaned by row lists, row-sparse Adam) and eval renders (full re-pack, all weight forms) -- and that the PSNR against
the teacher's images climbs.  Prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig  # noqa: E402
from pointnerf2studio_amd.ns_compat import RayBundle  # noqa: E402
from pointnerf2studio_amd.optim import PointRowAdam  # noqa: E402


def make_model(pts, weights, dev):
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    cfg = PointNerfConfig(ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=True,
                          collider_params={"near_plane": 2.0, "far_plane": 6.0}, hip_single_camera_bundles=True)
    model = PointNerf(cfg, point_state_dict=sd).to(dev)
    model.load_state_dict(weights, strict=False)
    if hasattr(model.collider, "reset_near_plane"):
        model.collider.reset_near_plane = False
    return model


def psnr(a, b):
    return float(-10.0 * math.log10(float(((a - b) ** 2).mean()) + 1e-20))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=500_000)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--side", type=int, default=200)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--eval-every", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H = W = args.side
    pts = synthetic.make_points(args.points, seed=1234)
    teacher = make_model(pts, synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1), dev)
    teacher.eval()
    teacher.neural_points.jitter = 0.0
    views = []
    for v in range(args.views):
        campos, camrot = synthetic.make_camera(45.0 * v + 20.0)
        dirs = synthetic.make_rays(H, W, campos, camrot).to(dev).reshape(H, W, 3)
        cam_bundle = RayBundle(origins=campos.to(dev)[None, None].expand(H, W, 3).contiguous(), directions=dirs,
                               metadata={"camrotc2w": camrot.to(dev)[None, None].expand(H, W, -1, -1).reshape(H, W, -1)})
        with torch.no_grad():
            target = teacher.get_outputs_for_camera_ray_bundle(cam_bundle)["coarse_raycolor"].reshape(H * W, 3).clone()
        views.append({"campos": campos.to(dev), "camrot": camrot.to(dev), "dirs": dirs.reshape(-1, 3), "bundle": cam_bundle,
                      "target": target})
    del teacher
    g = torch.Generator().manual_seed(5)
    spts = {k: v.clone() for k, v in pts.items()}
    spts["color"] = torch.rand(pts["color"].shape, generator=g)
    spts["embedding"] = pts["embedding"] + 0.1 * (torch.rand(pts["embedding"].shape, generator=g) - 0.5)
    model = make_model(spts, synthetic.make_weights(5, sigma_scale=300.0, bias_scale=0.1), dev)
    groups = model.get_param_groups()
    opts = [torch.optim.Adam(groups["fields"], lr=5e-4, eps=1e-8), PointRowAdam(groups["neural_points"], lr=2e-3, eps=1e-8)]
    scheds = [torch.optim.lr_scheduler.LambdaLR(o, lambda s: pow(0.1, s / 1000000)) for o in opts]
    callbacks = model.get_training_callbacks(None)

    def evaluate():
        model.eval()
        jit = model.neural_points.jitter
        model.neural_points.jitter = 0.0
        out = []
        with torch.no_grad():
            for v in views:
                rgb = model.get_outputs_for_camera_ray_bundle(v["bundle"])["coarse_raycolor"].reshape(-1, 3)
                out.append(psnr(rgb, v["target"]))
        model.neural_points.jitter = jit
        model.train()
        return out

    curve = [{"step": 0, "psnr_db": evaluate()}]
    torch.manual_seed(7)
    losses = []
    torch.cuda.synchronize()
    t0, t_train = time.perf_counter(), 0.0
    for step in range(1, args.steps + 1):
        ts = time.perf_counter()
        v = views[int(torch.randint(0, len(views), (1,)))]                    # one image per batch (random_image_idx)
        pick = torch.randint(0, H * W, (args.rays,), device=dev)
        b = RayBundle(origins=v["campos"][None].expand(args.rays, 3), directions=v["dirs"].index_select(0, pick),
                      metadata={"camrotc2w": v["camrot"]})
        for o in opts:
            o.zero_grad(set_to_none=True)
        out = model(b)
        loss = sum(model.get_loss_dict(out, {"image": v["target"].index_select(0, pick)}).values())
        loss.backward()
        for o, s in zip(opts, scheds):
            o.step()
            s.step()
        for cb in callbacks:
            cb.run_callback(step=step)
        losses.append(loss.detach())
        if step % args.eval_every == 0 or step == args.steps:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - ts
            curve.append({"step": step, "psnr_db": evaluate(), "loss": float(torch.stack(losses[-20:]).mean())})
            torch.cuda.synchronize()
        else:
            t_train += time.perf_counter() - ts
    torch.cuda.synchronize()
    ls = torch.stack(losses).cpu()
    res = {"points": args.points, "views": args.views, "image": [H, W], "rays_per_step": args.rays, "steps": args.steps,
           "psnr_curve": curve, "psnr_mean_db": [(c["step"], sum(c["psnr_db"]) / len(c["psnr_db"])) for c in curve],
           "loss_first_last": [float(ls[:10].mean()), float(ls[-10:].mean())], "all_finite": bool(torch.isfinite(ls).all()),
           "train_ms_per_step_wall": t_train / args.steps * 1e3, "wall_s": time.perf_counter() - t0,
           "host_reads": model.host_reads, "point_rows_ever_touched": opts[1].ever_touched(),
           "dense_sweeps": opts[1].dense_steps}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
