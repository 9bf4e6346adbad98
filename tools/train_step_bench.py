"""GPU box: time of one TRAINING step (forward + loss + backward) of the plugin model on the fused HIP path
(pnr_render + pnr_render_backward) and on the reference's op sequence under torch autograd, same model, same rays.

    python tools/train_step_bench.py [--points 6000000] [--rays 4096] [--steps 10]

The batch is what `ns-train pointnerf-original` draws: 4096 random pixels of one 800x800 camera
(studio_config.py:20-21 of the reference, one camera per bundle: studio_utils.py:152).  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig  # noqa: E402
from pointnerf2studio_amd.ns_compat import RayBundle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=6_000_000)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--skip-autograd", action="store_true")
    ap.add_argument("--no-sync", action="store_true", help="no host synchronisation between steps (events are read at the "
                    "end): the device never idles, as in a training loop whose host runs ahead")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    pts = synthetic.make_points(args.points)
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    cfg = PointNerfConfig(ranges=list(synthetic.CHAIR_RANGES), max_o=410000, enable_collider=False, hip_mlp_mode=args.mode)
    model = PointNerf(cfg, point_state_dict=sd).to(dev)
    # the autograd leg: the reference's op sequence under torch autograd (test infrastructure, hooked in)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from autograd_reference_path import get_outputs_autograd
    model.unfused_outputs_fn = get_outputs_autograd
    model.load_state_dict(synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1), strict=False)
    model.train()
    campos, camrot = synthetic.make_camera(35.0, 30.0)
    dirs_all = synthetic.make_rays(800, 800, campos, camrot)
    g = torch.Generator().manual_seed(11)
    pick = torch.randperm(dirs_all.shape[0], generator=g)[:args.rays]
    dirs = dirs_all[pick].contiguous()
    R = dirs.shape[0]
    bundle = RayBundle(origins=campos[None].expand(R, 3).to(dev), directions=dirs.to(dev),
                       nears=torch.full((R, 1), 2.0, device=dev), fars=torch.full((R, 1), 6.0, device=dev),
                       metadata={"camrotc2w": camrot.reshape(1, 9).expand(R, 9).to(dev)})
    image = torch.rand(R, 3, generator=g).to(dev)

    def run(fused):
        model.config.hip_fused_training = fused
        fw, bw = [], []
        counters = None
        events = []
        for it in range(args.warmup + args.steps):
            model.zero_grad(set_to_none=True)
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            if not args.no_sync:
                torch.cuda.synchronize()
            e[0].record()
            out = model(bundle)
            loss = sum(model.get_loss_dict(out, {"image": image}).values())
            e[1].record()
            loss.backward()
            e[2].record()
            if not args.no_sync:
                torch.cuda.synchronize()
            if it >= args.warmup:
                events.append(e)
            if fused:
                counters = model._renderer_train.last_counters
        torch.cuda.synchronize()
        for e in events:
            fw.append(e[0].elapsed_time(e[1]))
            bw.append(e[1].elapsed_time(e[2]))
        fw.sort()
        bw.sort()
        return {"forward_ms": fw[len(fw) // 2], "backward_ms": bw[len(bw) // 2],
                "step_ms": fw[len(fw) // 2] + bw[len(bw) // 2], "loss": loss.item(), "counters": counters}

    res = {"points": args.points, "rays": R, "mode_forward": args.mode}
    t0 = time.time()
    res["fused"] = run(True)
    res["fused"]["rays_per_sec"] = R / (res["fused"]["step_ms"] * 1e-3)
    if not args.skip_autograd:
        res["autograd"] = run(False)
        res["autograd"]["rays_per_sec"] = R / (res["autograd"]["step_ms"] * 1e-3)
        res["speedup"] = res["autograd"]["step_ms"] / res["fused"]["step_ms"]
    res["wall_s"] = time.time() - t0
    print(json.dumps(res))


if __name__ == "__main__":
    main()
