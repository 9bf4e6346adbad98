"""GPU box: the small-batch calls of the C ABI (dozens of dependent launches per call) captured in a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture /
hipGraphLaunch) and replayed, next to the same calls issued launch by launch:

  * an eval frame of BASELINE cfg[0] (50 k points, 64 x 64 rays: about twenty launches, 0.4 ms) -- pnr_render_views;
  * a 4096-ray TRAINING step over the 6 M-point cloud (taped pnr_render_views + pnr_render_backward accumulating into
    persistent point-gradient buffers + pnr_render_touched + pnr_point_grads_clear: about fifty launches, 2 ms).

    python tools/graph_replay.py [--points 6000000] [--iters 200]

Every entry of the render path takes its sizes from device memory and never returns to the host, so a call IS capturable;
the replayed results are compared with the directly launched ones bit for bit.  A captured call bakes its host arguments
(camera, jitter seed) into the graph: a replay repeats THAT render.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.renderer import (RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters)  # noqa: E402

VSCALE, KSIZE = [2, 2, 2], [3, 3, 3]
STAGE_LOG = os.environ.get("PNR_GRAPH_STAGE_LOG")


def stage(msg):
    """progress line after a device synchronisation (a GPU fault ends the process: the log then names the stage it was in)"""
    if STAGE_LOG:
        torch.cuda.synchronize()
        with open(STAGE_LOG, "a") as f:
            f.write(msg + "\n")


def timed(fn, iters, what=""):
    for _ in range(5):
        fn()
        stage(f"timed {what}: one call, synchronised")
    for _ in range(5):
        fn()
    stage(f"timed {what}: five calls back to back")
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters * 1e3
    stage(f"timed {what}: {iters} calls back to back")
    return dt


def capture(fn):
    """fn on a side stream a few times (allocator warm-up, as torch's graph notes ask), then captured"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    stage("capture: warm-up on a side stream done")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    stage("capture: graph instantiated")
    return g, keep


def eval_frame(dev, iters):
    c = dict(synthetic.SCENE_CONFIGS["cfg0_chair_50k"])
    H, W, SR, K = c["H"], c["W"], c["SR"], c["K"]
    points = synthetic.make_scene_points(c, seed=1234)
    weights = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    campos, camrot = synthetic.make_scene_camera(c, 0)
    dirs = synthetic.make_rays(H, W, campos, camrot, c["angle_x"]).to(dev)
    xyz = points["xyz"].to(dev)
    hyp = grid_hyperparameters(xyz, [c["vsize"]] * 3, VSCALE, KSIZE, c["ranges"])
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, KSIZE, KSIZE, c["P"], c["max_o"], True)
    scene.pack_points(xyz, *(points[k].to(dev) for k in ("embedding", "conf", "dir", "color")))
    wh = WeightsHIP()
    wh.pack(weights, points["Rw2c"], dev)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * c["vsize"], vsize_z=c["vsize"], jitter=0.3, seed=7)
    out = rnd.render(dirs, campos, camrot, c["near"], c["far"])
    cap = int(out["counters"]["samples_selected"] * 1.05) + 4096

    def direct():
        return rnd.render(dirs, campos, camrot, c["near"], c["far"], cap_samples=cap, sync_counters=False, out=out)
    stage("eval: first render done")
    direct()
    torch.cuda.synchronize()
    stage("eval: direct render done")
    want = {k: out[k].clone() for k in ("rgb", "depth", "acc", "ray_mask", "counters_dev")}
    g, _ = capture(direct)
    for k in want:
        out[k].fill_(0)
    g.replay()
    torch.cuda.synchronize()
    stage("eval: first replay done")
    same = all(torch.equal(out[k], want[k]) for k in want)
    return {"workload": f"cfg0_chair_50k eval frame, {H}x{W} rays, jitter 0.3", "launches_ms": timed(direct, iters, "eval launches"),
            "graph_replay_ms": timed(g.replay, iters, "eval replay"), "replay_bit_identical": bool(same),
            "rays_kept": int(want["counters_dev"][1])}


def training_step(dev, n_points, n_rays, iters):
    pts = synthetic.make_points(n_points, seed=1234)
    weights = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    xyz = pts["xyz"].to(dev)
    hyp = grid_hyperparameters(xyz, [0.004] * 3, VSCALE, KSIZE, synthetic.CHAIR_RANGES)
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, KSIZE, KSIZE, 12, 410000, True)
    live = [pts[k].to(dev).contiguous() for k in ("embedding", "conf", "dir", "color")]
    scene.pack_points(xyz, *live)
    wh = WeightsHIP()
    wh.pack(weights, pts["Rw2c"], dev)
    w_dev = {k: v.to(dev).contiguous() for k, v in weights.items()}
    rnd = RendererHIP(scene, wh, SR=80, K=8, D=400, radius_limit=0.016, vsize_z=0.004, eval_clamp=False, jitter=0.3, seed=1,
                      tape=True)
    campos, camrot = synthetic.make_camera(35.0, 30.0)
    full = synthetic.make_rays(800, 800, campos, camrot)
    gen = torch.Generator().manual_seed(11)
    dirs = full[torch.randperm(full.shape[0], generator=gen)[:n_rays]].contiguous().to(dev)
    g_rgb = torch.randn(n_rays, 3, generator=gen).to(dev)
    out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
    cap = int(out["counters"]["samples_selected"] * 1.25) + 4096
    N = n_points
    into = {"embedding": torch.zeros(N * 32, device=dev), "color": torch.zeros(N * 3, device=dev),
            "dir": torch.zeros(N * 3, device=dev)}
    rnd.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap, sync_counters=False, out=out)
    index, count = rnd.touched()
    keep = {}

    def direct():
        rnd.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap, sync_counters=False, out=out)
        g = rnd.backward(g_rgb, w_dev, N, into=into)
        rnd.touched(index, count)
        # (a training loop's optimiser reads the rows here; then they are zeroed for the next step)
        keep["emb_rows"] = into["embedding"].view(N, 32).index_select(0, index[:256].long())
        rnd.clear_point_grads(into["embedding"], into["color"], into["dir"], N, index, count)
        keep["g"] = g
        return g
    stage("train: set-up renders done")
    direct()
    torch.cuda.synchronize()
    stage("train: direct step done")
    want = {k: v.clone() for k, v in keep["g"].items()}
    want_rows = keep["emb_rows"].clone()
    g, gres = capture(direct)
    for v in gres.values():
        v.fill_(0)
    g.replay()
    torch.cuda.synchronize()
    stage("train: first replay done")
    same = all(torch.equal(gres[k], want[k]) for k in want) and torch.equal(keep["emb_rows"], want_rows)
    clean = bool((into["embedding"] == 0).all() and (into["color"] == 0).all() and (into["dir"] == 0).all())
    return {"workload": f"training step, {n_rays} rays of one 800x800 camera over {n_points} points, taped render + backward + "
                        f"touched-row list + row clear", "launches_ms": timed(direct, iters, "train launches"),
            "graph_replay_ms": timed(g.replay, iters, "train replay"), "replay_bit_identical": bool(same),
            "point_grad_buffers_clean_after_step": clean, "touched_points": int(count.item())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=6_000_000)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"eval_frame": eval_frame(dev, args.iters), "training_step": training_step(dev, args.points, args.rays, args.iters)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
