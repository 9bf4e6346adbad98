"""GPU box, beside tools/bin/ub_clock_probe (started first, as another process): the phases of the 65 536-ray training step
run ONE AT A TIME for a couple of seconds each -- untaped render, taped render, backward after a taped render, the whole
step -- and the host times of every phase, so that the probe's windows can be read per phase.

    tools/bin/ub_clock_probe 20 2 > gpurun_out/probe_phases.txt &  sleep 0.5
    python tools/clock_probe_phases.py --probe gpurun_out/probe_phases.txt
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--probe", default=None, help="the probe's output file (read after it has ended)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N = 6_000_000
    pts = synthetic.make_points(N, seed=1234)
    weights = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    xyz = pts["xyz"].to(dev)
    hyp = grid_hyperparameters(xyz, [0.004] * 3, [2, 2, 2], [3, 3, 3], synthetic.CHAIR_RANGES)
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, [3, 3, 3], [3, 3, 3], 12, 410000, True)
    scene.pack_points(xyz, *(pts[k].to(dev) for k in ("embedding", "conf", "dir", "color")))
    wh = WeightsHIP()
    wh.pack(weights, pts["Rw2c"], dev)
    w_dev = {k: v.to(dev).contiguous() for k, v in weights.items()}
    campos, camrot = synthetic.make_camera(35.0, 30.0)
    full = synthetic.make_rays(800, 800, campos, camrot)
    gen = torch.Generator().manual_seed(11)
    dirs = full[torch.randperm(full.shape[0], generator=gen)[:args.rays]].contiguous().to(dev)
    g_rgb = torch.randn(args.rays, 3, generator=gen).to(dev)
    into = {"embedding": torch.zeros(N * 32, device=dev), "color": torch.zeros(N * 3, device=dev),
            "dir": torch.zeros(N * 3, device=dev)}

    def renderer(tape):
        r = RendererHIP(scene, wh, SR=80, K=8, D=400, radius_limit=0.016, vsize_z=0.004, eval_clamp=False, jitter=0.3, seed=1,
                        tape=tape)
        o = r.render(dirs, campos, camrot, 2.0, 6.0)
        return r, o, r.cap_samples
    r0, o0, cap0 = renderer(False)
    r1, o1, cap1 = renderer(True)
    index, count = r1.touched()

    def render_untaped():
        r0.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap0, sync_counters=False, out=o0)

    def render_taped():
        r1.render(dirs, campos, camrot, 2.0, 6.0, cap_samples=cap1, sync_counters=False, out=o1)

    def backward():          # (again and again on the tape of the last taped render: a backward does not consume it)
        r1.backward(g_rgb, w_dev, N, into=into)
        r1.clear_point_grads(into["embedding"], into["color"], into["dir"], N, index, count)

    def step():
        render_taped()
        backward()
    phases = []
    for name, fn in (("render, untaped", render_untaped), ("render, taped", render_taped), ("backward (taped)", backward),
                     ("whole step", step)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        time.sleep(0.4)
        t0 = time.time()
        n = 0
        while time.time() - t0 < args.seconds:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
            n += 8
        t1 = time.time()
        phases.append({"phase": name, "t0": t0, "t1": t1, "calls": n, "ms_per_call": (t1 - t0) / n * 1e3})
        time.sleep(0.4)
    if args.probe:
        time.sleep(0.5)
        deadline = time.time() + 30
        while time.time() < deadline:      # the probe ends by itself; its file is complete when the last line is there
            txt = open(args.probe).read()
            if txt.count("\n") > 10 and not txt.endswith(":"):
                rows = [l.split() for l in txt.splitlines() if l and l[0] != "#"]
                t_launch = float([l for l in txt.splitlines() if l.startswith("# t0")][0].split()[2])
                if rows and float(rows[-1][1]) / 1e3 + t_launch >= phases[-1]["t1"]:
                    break
            time.sleep(0.5)
        for ph in phases:
            v = sorted(float(r[2]) for r in rows if ph["t0"] + 0.1 <= t_launch + float(r[1]) / 1e3 <= ph["t1"] - 0.1)
            if v:
                ph["clock_mhz"] = {"windows": len(v), "p10": v[len(v) // 10], "median": v[len(v) // 2], "p90": v[9 * len(v) // 10]}
    for ph in phases:
        ph.pop("t0"), ph.pop("t1")
    print(json.dumps({"rays": args.rays, "phases": phases}))


if __name__ == "__main__":
    main()
