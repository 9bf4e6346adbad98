"""GPU box, one-off soak: pnr_render_backward (fp32, taped and recomputing) against torch autograd through the CPU oracle over
RANDOM configurations -- K 1..16, SR, P, jitter, clamp / no clamp, cloud size, camera.  Reports per case the worst relative
error (of a tensor's largest magnitude) and the relative L2 over all gradient tensors; a LeakyReLU unit within rounding of
its kink may move single entries by per cent (tools/kink_sweep.py), a wrong kernel moves whole tensors.
    python tools/backward_fuzz.py [--cases 24]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import pnr_oracle as O  # noqa: E402
from helpers import build_hip, camera_rays, oracle_cfg, small_scene  # noqa: E402
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER, RendererHIP  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    args = ap.parse_args()
    O.build_c_oracle()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(20261005)
    worst = []
    for i in range(args.cases):
        K = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10, 12, 13, 16]))
        SR = int(rng.choice([4, 16, 24, 32, 80]))
        P = int(rng.choice([3, 12, 26]))
        N = int(rng.choice([20000, 60000, 150000]))
        jitter = float(rng.choice([0.0, 0.3]))
        training = bool(rng.rand() < 0.6)
        H, W = int(rng.choice([12, 20, 28])), int(rng.choice([12, 20, 28]))
        az, el = float(rng.uniform(0, 360)), float(rng.uniform(-10, 50))
        sigma = float(rng.choice([30.0, 300.0]))
        pts = small_scene(N, seed=500 + i)
        cfg = oracle_cfg(O, SR=SR, K=K, P=P)
        w = synthetic.make_weights(i, sigma_scale=sigma, bias_scale=0.1)
        campos, camrot, dirs = camera_rays(H, W, az=az, el=el)
        R = dirs.shape[0]
        G = torch.randn(R, 3, generator=torch.Generator().manual_seed(i))
        u = O.jitter_uniforms(R, cfg.z_depth_dim, seed=3 + i) if jitter > 0 else None
        pts_g = dict(pts)
        for k in ("embedding", "color", "dir"):
            pts_g[k] = pts[k].clone().requires_grad_(True)
        w_g = {k: v.clone().requires_grad_(True) for k, v in w.items()}
        ref = O.render(pts_g, w_g, cfg, campos[None].expand(R, 3), dirs, 2.0, 6.0, camrot, jitter=jitter, u=u, training=training)
        if ref["stats"]["valid_pairs"] < 50:
            continue
        (ref["coarse_raycolor"] * G).sum().backward()
        want = {k: pts_g[k].grad.reshape(pts_g[k].shape[-2], -1) for k in ("embedding", "color", "dir")}
        want.update({k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in w_g.items()})
        scene, wh, hyp, info = build_hip(pts, cfg, dev, weights=w)
        for tape in (False, True):
            rnd = RendererHIP(scene, wh, SR=SR, K=K, eval_clamp=not training, jitter=jitter, seed=3 + i, tape=tape)
            rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)
            got = rnd.backward(G.to(dev), {k: v.to(dev) for k, v in w.items()}, N)
            mx, l2, who = 0.0, 0.0, ""
            for k in ["embedding", "color", "dir"] + [n + s for n in MLP_TENSOR_ORDER for s in (".weight", ".bias")]:
                a, b = got[k].cpu(), want[k]
                sc = b.abs().max().item()
                if sc == 0:
                    continue
                e = (a - b).abs().max().item() / sc
                n2 = ((a - b).double().norm() / b.double().norm()).item()
                if e > mx:
                    mx, who = e, k
                l2 = max(l2, n2)
            worst.append({"case": i, "K": K, "SR": SR, "P": P, "N": N, "jitter": jitter, "training": training, "tape": tape,
                          "pairs": int(ref["stats"]["valid_pairs"]), "max_rel": mx, "max_rel_tensor": who, "max_l2": l2,
                          "image_err": (got["rgb"].cpu() - ref["coarse_raycolor"]).abs().max().item()})
            print(json.dumps(worst[-1]), flush=True)
    bad = [x for x in worst if x["max_l2"] > 1e-2 or x["image_err"] > 1e-4]
    print(json.dumps({"cases_run": len(worst), "max_rel_overall": max(x["max_rel"] for x in worst),
                      "max_l2_overall": max(x["max_l2"] for x in worst), "suspicious": bad}))


if __name__ == "__main__":
    main()
