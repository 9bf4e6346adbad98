"""Diagnostic: per-phase s_memtime shares of k_shade_pairs_bf16 (needs a -DPNR_STAMPS=1 build).
Usage on the GPU box: PNR_LIB=$PWD/pointnerf2studio_amd/_abl/libpnr_stamps.so python tools/stamps_shade.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters

dev = torch.device("cuda:0")
cfg = synthetic.SCENE_CONFIGS["cfg1_chair_6m"]
pts = synthetic.make_points(cfg["N"], seed=1234)
w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
xyz = pts["xyz"].to(dev)
hyp = grid_hyperparameters(xyz, [0.004] * 3, [2, 2, 2], [3, 3, 3], cfg["ranges"])
scene = SceneHIP()
scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, [3, 3, 3], [3, 3, 3], cfg["P"], cfg["max_o"])
scene.pack_points(xyz, pts["embedding"].to(dev), pts["conf"].to(dev), pts["dir"].to(dev), pts["color"].to(dev))
wh = WeightsHIP(); wh.pack(w, pts["Rw2c"], dev)
rnd = RendererHIP(scene, wh, precision="bf16x3")
campos, camrot = synthetic.make_camera(65.0)
dirs = synthetic.make_rays(800, 800, campos, camrot).to(dev)
out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
cap = rnd.cap_samples
out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
torch.cuda.synchronize()
# locate smp_sigma inside the workspace: it follows scan_temp; recompute via the taps' neighbours is overkill:
# the stamp area is the last 8192 floats of the sigma buffer = [ws_sigma + cap - 8192, ws_sigma + cap)
import ctypes as C
from pointnerf2studio_amd import _lib
lib = _lib.load()
total = lib.pnr_render_workspace_bytes(dirs.shape[0], cap, 8)
agg_bytes = (cap + 32) * 256 * 4
sigma_bytes = (cap * 4 + 255) // 256 * 256
sigma_off = total - ((agg_bytes + 255) // 256 * 256) - sigma_bytes
raw = rnd._ws[sigma_off + (cap - 8192) * 4: sigma_off + cap * 4].view(torch.int64).view(-1, 8)[:256].cpu().double()
raw = raw[raw[:, 7] > 0]
per_tile = raw[:, :7] / raw[:, 7:8]
names = ["prologue(total)", "layer1", "layer2", "layer3", "layer4", "epilogue", "  of prologue: camera + gather issue"]
m = per_tile.mean(0)
tot = m[:6].sum()
print(f"waves sampled {raw.shape[0]}, tiles/wave {raw[:,7].mean():.1f}, s_memtime ticks per tile {tot:.0f}")
for n, v in zip(names, m.tolist()):
    print(f"  {n:22s} {v:9.0f}  {100 * v / tot:5.1f} %")
