"""Not a test (not collected): the numerical experiment behind the choice of THREE bf16 products per fp32 product.
The CPU oracle renders small scenes with F.linear replaced by emulations of cheaper split schemes; printed is the
maximum absolute RGB / depth difference to the exact fp32 render.  Result (2026-10, four scenes):
  bf16x3 (wh.xh + wh.xl + wl.xh)            3e-6 .. 8e-6      <- what PNR_PRECISION_BF16X3 does (measured on GPU: 7e-6)
  fp16, weights exact, activations rounded   1e-4 .. 3e-4      two products: over the 1e-4 budget
  fp16, activations exact, weights rounded   2e-4 .. 3e-4      two products: over the 1e-4 budget
  bf16, weights exact, activations rounded   8e-4 .. 3e-3
Run: python tests/precision_experiment.py   (CPU, a few minutes)"""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import pnr_oracle as O
import torch.nn.functional as F
from helpers import camera_rays, oracle_cfg, small_scene
from pointnerf2studio_amd import synthetic
torch.set_num_threads(8)
def run(mode, pts, w, cfg, campos, camrot, dirs):
    orig_lin = F.linear
    def lin(x, W, b=None):
        if mode == "exact": return orig_lin(x, W, b)
        if mode == "fp16act":   # W (hi+lo fp16 = ~22 bits) x fp16(x): two products
            Wq = W.half().float() + (W - W.half().float()).half().float()
            return orig_lin(x.half().float(), Wq, b)
        if mode == "fp16wt":    # fp16(W) x (xh + xl): two products, weights rounded
            xq = x.half().float() + (x - x.half().float()).half().float()
            return orig_lin(xq, W.half().float(), b)
        if mode == "bf16x3":
            xh = x.bfloat16().float(); xl = (x - xh).bfloat16().float()
            Wh = W.bfloat16().float(); Wl = (W - Wh).bfloat16().float()
            y = orig_lin(xh, Wh) + orig_lin(xl, Wh) + orig_lin(xh, Wl)
            return y + b if b is not None else y
        if mode == "bf16x2act":
            xh = x.bfloat16().float()
            Wh = W.bfloat16().float(); Wl = (W - Wh).bfloat16().float()
            y = orig_lin(xh, Wh) + orig_lin(xh, Wl)
            return y + b if b is not None else y
    F.linear = lin
    try:
        return O.render(pts, w, cfg, campos[None].expand(dirs.shape[0], 3), dirs, 2.0, 6.0, camrot)
    finally:
        F.linear = orig_lin
for N, az, ss in ((60000, 35.0, 300.0), (200000, 120.0, 300.0), (60000, 200.0, 30.0), (120000, 300.0, 1000.0)):
    pts = small_scene(N); cfg = oracle_cfg(O)
    w = synthetic.make_weights(0, sigma_scale=ss, bias_scale=0.1)
    campos, camrot, dirs = camera_rays(32, 32, az=az)
    ref = run("exact", pts, w, cfg, campos, camrot, dirs)
    for mode in ("bf16x3", "fp16act", "fp16wt", "bf16x2act"):
        o = run(mode, pts, w, cfg, campos, camrot, dirs)
        e = (o["coarse_raycolor"] - ref["coarse_raycolor"]).abs().max().item()
        d = (o["depth"] - ref["depth"]).abs().max().item()
        print(f"N={N} az={az} sigma_scale={ss}: {mode:10s} max|dRGB| {e:.2e} max|ddepth| {d:.2e}")
