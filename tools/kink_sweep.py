"""GPU box: the two forms of the training step's backward (the backward recomputing the MLP chain itself / the render
writing the activation tape, pnr_render_opts_t.d_tape) against torch autograd through the CPU oracle over a few views,
jitter seeds and cameras.  Three float32 evaluations of the same network in three summation orders: almost every entry
agrees to 1e-6, but a LeakyReLU unit whose pre-activation is within rounding of zero falls on either side of the kink
(derivative 1 or 0.1), and with the bench's density scale (300) a single such unit can carry per cent of a gradient
tensor's largest entry.  Which of the three flips is a coin toss per case -- measured: recompute 1.4e-2 / taped 5.7e-5 in
one case, 4.6e-5 / 9.5e-4 in another, both 2.2e-3 in a third (there the oracle is the odd one out).
    gpurun -- python tools/kink_sweep.py"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import pnr_oracle as O
from helpers import build_hip, camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER, RendererHIP
dev = torch.device("cuda:0")
K, SR, P = 8, 80, 12
pts = small_scene(80000); cfg = oracle_cfg(O, SR=SR, K=K, P=P)
w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
scene, wh, hyp, info = build_hip(pts, cfg, dev, weights=w)
N = pts["xyz"].shape[0]
wd = {k: v.to(dev) for k, v in w.items()}
for jitter, seed, az in [(0.3, 5, 75.0), (0.0, 5, 75.0), (0.3, 6, 75.0), (0.3, 7, 200.0), (0.0, 5, 200.0), (0.3, 8, 120.0)]:
  campos, camrot, dirs = camera_rays(28, 36, az=az)
  R = dirs.shape[0]
  G = torch.randn(R, 3, generator=torch.Generator().manual_seed(3))
  u = O.jitter_uniforms(R, 400, seed)
  pts_g = dict(pts); pts_g["embedding"] = pts["embedding"].clone().requires_grad_(True)
  w_g = {k: v.clone().requires_grad_(True) for k, v in w.items()}
  ref = O.render(pts_g, w_g, cfg, campos[None].expand(R, 3), dirs, 2.0, 6.0, camrot, jitter=jitter, u=u, training=True)
  (ref["coarse_raycolor"] * G).sum().backward()
  res = {}
  for tape in (False, True):
    rnd = RendererHIP(scene, wh, SR=SR, K=K, eval_clamp=False, tape=tape, jitter=jitter, seed=seed)
    rnd.render(dirs.to(dev), campos, camrot, 2.0, 6.0)
    res[tape] = {k: v.cpu() for k, v in rnd.backward(G.to(dev), wd, N).items()}
  want = {"embedding": pts_g["embedding"].grad.reshape(-1, 32)}
  want.update({k: v.grad for k, v in w_g.items()})
  print("jitter", jitter, "seed", seed, "az", az)
  for k in ["embedding", "mlp_base.layers.1.weight", "mlp_head.layers.1.weight"]:
    sc = want[k].abs().max().item()
    e0 = (res[False][k] - want[k]).abs().max().item() / sc
    e1 = (res[True][k] - want[k]).abs().max().item() / sc
    print(f"   {k:30s} recompute-vs-oracle {e0:.2e}   taped-vs-oracle {e1:.2e}")
