"""Diagnostic: the other BASELINE.json configurations at FULL size on one GPU (no oracle at these sizes: properties
only) -- DTU-style (10 M points, 1600x1200, K = 8) and ScanNet-style (20 M points, 1296x968, K = 12, SR = 24, P = 26,
camera inside the cloud).  Checks: no capacity overflow, accumulated opacity in [0, 1], background rays exactly white,
the two arithmetic modes agree to 1e-4, a re-render is bitwise identical.  Prints stage times.
Usage on the GPU box: python tools/scale_check.py [dtu|scannet]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from pointnerf2studio_amd import _lib, synthetic  # noqa: E402
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "scannet"
dev = torch.device("cuda:0")
if which == "dtu":
    N, H, W, K, SR, P, vs, ranges = 10_000_000, 1200, 1600, 8, 80, 12, 0.004, synthetic.CHAIR_RANGES
    pts = synthetic.make_points(N, ranges=ranges)
    campos, camrot = synthetic.make_camera(40.0)
    dirs = synthetic.make_rays(H, W, campos, camrot)
    near, far, max_o = 2.0, 6.0, 410000
else:
    N, H, W, K, SR, P, vs = 20_000_000, 968, 1296, 12, 24, 26, 0.008
    ranges = [-0.5, -0.5, -0.5, 8.5, 6.5, 3.5]
    pts = synthetic.make_room_points(N)
    campos, camrot = synthetic.make_inside_camera([4.0, 3.0, 1.5], yaw_deg=35.0, pitch_deg=-10.0)
    dirs = synthetic.make_rays(H, W, campos, camrot, camera_angle_x=1.0)
    near, far, max_o = 0.1, 8.0, 1000000
w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
xyz = pts["xyz"].to(dev)
hyp = grid_hyperparameters(xyz, [vs] * 3, [2, 2, 2], [3, 3, 3], ranges)
scene = SceneHIP()
t0 = time.time()
info = scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, [3, 3, 3], [3, 3, 3], P, max_o, True)
torch.cuda.synchronize()
print(which, "scene build %.1f ms" % ((time.time() - t0) * 1e3), info)
scene.pack_points(xyz, pts["embedding"].to(dev), pts["conf"].to(dev), pts["dir"].to(dev), pts["color"].to(dev))
wh = WeightsHIP()
wh.pack(w, pts["Rw2c"], dev)
d = dirs.to(dev)
lib = _lib.load()
res = {}
for mode in ("bf16x3", "fp32"):
    rnd = RendererHIP(scene, wh, SR=SR, K=K, D=400, radius_limit=4 * vs, vsize_z=vs, precision=mode)
    o = rnd.render(d, campos, camrot, near, far)          # sizes the workspace
    rgb0, mask0 = o["rgb"].clone(), o["ray_mask"].clone()
    lib.pnr_profile_enable(1)
    o = rnd.render(d, campos, camrot, near, far)
    torch.cuda.synchronize()
    ms = (C.c_float * _lib.NUM_STAGES)()
    _lib.check(lib.pnr_profile_read(int(lib.pnr_profile_calls()) - 1, C.byref(ms)), "profile")
    lib.pnr_profile_enable(0)
    assert torch.equal(o["rgb"], rgb0) and torch.equal(o["ray_mask"], mask0), "re-render differs"
    assert o["counters"]["overflow"] == 0
    assert float(o["acc"].min()) >= 0.0 and float(o["acc"].max()) <= 1.0 + 1e-5
    assert torch.all(o["rgb"][o["ray_mask"] == 0] == 1.0)
    total = sum(ms)
    print(f"{mode}: {d.shape[0]} rays, {total:.2f} ms = {d.shape[0] / total / 1e3:.2f} M rays/s;",
          {n: round(ms[i], 3) for i, n in enumerate(_lib.STAGE_NAMES)},
          {k: o["counters"][k] for k in ("rays_kept", "samples_valid", "pairs_valid", "points_unique")},
          "workspace %.2f GB" % (rnd._ws.numel() / 1e9))
    res[mode] = o["rgb"].clone()
print("max |rgb(bf16x3) - rgb(fp32)| = %.2e" % (res["bf16x3"] - res["fp32"]).abs().max().item())
assert (res["bf16x3"] - res["fp32"]).abs().max().item() <= 1e-4
print("ok")
