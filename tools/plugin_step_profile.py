#!/usr/bin/env python3
"""Where a plugin training step goes (gpurun -- python tools/plugin_step_profile.py): torch profiler table of 10 steps of
PointNerf.forward + get_loss_dict + backward (+ Adam) on the bench's 6 M-point scene at 4096 rays."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointnerf2studio_amd import synthetic  # noqa: E402
from pointnerf2studio_amd.model import PointNerf, PointNerfConfig  # noqa: E402
from pointnerf2studio_amd.ns_compat import RayBundle  # noqa: E402

dev = torch.device("cuda:0")
cfgd = dict(synthetic.SCENE_CONFIGS["cfg1_chair_6m"])
if len(sys.argv) > 1:
    cfgd["N"] = int(sys.argv[1])
points = synthetic.make_scene_points(cfgd, seed=1234)
weights = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
sd = {"neural_points.xyz": points["xyz"], "neural_points.points_embeding": points["embedding"],
      "neural_points.points_conf": points["conf"], "neural_points.points_dir": points["dir"],
      "neural_points.points_color": points["color"], "neural_points.Rw2c": points["Rw2c"]}
cfg = PointNerfConfig(ranges=list(cfgd["ranges"]), max_o=cfgd["max_o"], SR=cfgd["SR"], K=cfgd["K"], P=cfgd["P"],
                      vsize=[cfgd["vsize"]] * 3, enable_collider=False)
model = PointNerf(cfg, point_state_dict=sd).to(dev)
model.load_state_dict(weights, strict=False)
model.train()
H, W = cfgd["H"], cfgd["W"]
campos, camrot = synthetic.make_scene_camera(cfgd, 0)
full = synthetic.make_rays(H, W, campos, camrot, cfgd["angle_x"]).to(dev)
opt = torch.optim.Adam([{"params": g} for g in model.get_param_groups().values()], lr=1e-4)
gen = torch.Generator().manual_seed(12)
n = 4096


def step(adam):
    pick = torch.randint(0, full.shape[0], (n,), device=dev)
    b = RayBundle(origins=campos.to(dev)[None].expand(n, 3), directions=full.index_select(0, pick),
                  nears=torch.full((n, 1), cfgd["near"], device=dev), fars=torch.full((n, 1), cfgd["far"], device=dev),
                  metadata={"camrotc2w": camrot.to(dev)})
    opt.zero_grad(set_to_none=True)
    out = model(b)
    sum(model.get_loss_dict(out, {"image": torch.rand((n, 3), device=dev)}).values()).backward()
    if adam:
        opt.step()
    for cb in model.get_training_callbacks(None):
        cb.run_callback(step=0)


for adam in (False, True):
    for _ in range(5):
        step(adam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step(adam)
    torch.cuda.synchronize()
    print(f"adam={adam}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10):
        step(False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=70))
