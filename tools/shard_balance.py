"""Diagnostic (GPU box): load balance of tile-to-rank assignments.  Renders the eight views of cfg 1 once, takes the
valid (sample, neighbour) pairs of every ray from the workspace and sums them per rank for several assignments of the
16x16 tiles (the cost of a rank is proportional to its pairs).   python tools/shard_balance.py [world]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, View, grid_hyperparameters

dev = torch.device("cuda:0")
cfg = synthetic.SCENE_CONFIGS["cfg1_chair_6m"]
H, W = cfg["H"], cfg["W"]
pts = synthetic.make_scene_points(cfg, seed=1234)
wts = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
xyz = pts["xyz"].to(dev)
VS = [cfg["vsize"]] * 3
hyp = grid_hyperparameters(xyz, VS, [2, 2, 2], [3, 3, 3], cfg["ranges"])
scene = SceneHIP()
scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, [3, 3, 3], [3, 3, 3], cfg["P"], cfg["max_o"], True)
scene.pack_points(xyz, pts["embedding"].to(dev), pts["conf"].to(dev), pts["dir"].to(dev), pts["color"].to(dev))
wh = WeightsHIP(); wh.pack(wts, pts["Rw2c"], dev)
rnd = RendererHIP(scene, wh, SR=cfg["SR"], K=cfg["K"], D=400, radius_limit=4 * VS[0], vsize_z=VS[2], jitter=0.3, seed=7)
cost = []   # per view: pairs per pixel [H*W]
for az in range(8):
    cp, cr = synthetic.make_scene_camera(cfg, az)
    out = rnd.render_camera([View.from_angle(cp, cr, H, W, cfg["angle_x"], cfg["near"], cfg["far"])], H, W)
    S = int(out["counters"]["samples_selected"])
    t = rnd.taps(H * W)
    pairs = (t["smp_pidx"][:S] >= 0).sum(1).float()
    c = torch.zeros(H * W, device=dev).index_add_(0, t["smp_ray"][:S].long(), pairs)
    cost.append(c.cpu().view(H, W))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nty, ntx = (H + T - 1) // T, (W + T - 1) // T
tile_cost = torch.stack([c.view(nty, T, ntx, T).sum((1, 3)) for c in cost])   # [8, nty, ntx]
ty, tx = torch.meshgrid(torch.arange(nty), torch.arange(ntx), indexing="ij")
schemes = {
    "idx % world (shipped)": (ty * ntx + tx) % world,
    "(tx + 3 ty) % world": (tx + 3 * ty) % world,
    "(tx + 5 ty) % world": (tx + 5 * ty) % world,
    "(tx + ty) % world": (tx + ty) % world,
    "(tx + 3 ty + (ty // world)) % world": (tx + 3 * ty + ty // world) % world,
}
def rotated(owner):
    worst = []
    for s0 in range(8):
        vs = [(s0 * world + i) % 8 for i in range(world)]
        per_rank = torch.stack([sum(tile_cost[v][(owner + i) % world == r].sum() for i, v in enumerate(vs)) for r in range(world)])
        worst.append((per_rank.max() / per_rank.mean()).item())
    return worst
print(f"tile {T}: idx % world, owner rotated by the view's position in the step: step max/mean " +
      " ".join(f"{x:.3f}" for x in sorted(set(round(w, 3) for w in rotated((ty * ntx + tx) % world)))))
for name, owner in schemes.items():
    # weak-scaling step: `world` views per step (views s*world + i), each rank renders its tiles of all of them
    worst = []
    for s0 in range(8):
        vs = [(s0 * world + i) % 8 for i in range(world)]
        per_rank = torch.stack([sum(tile_cost[v][owner == r].sum() for v in vs) for r in range(world)])
        worst.append((per_rank.max() / per_rank.mean()).item())
    per_view = []
    for v in range(8):
        pr = torch.stack([tile_cost[v][owner == r].sum() for r in range(world)])
        per_view.append((pr.max() / pr.mean()).item())
    print(f"{name:38s} step max/mean: " + " ".join(f"{x:.3f}" for x in sorted(set(round(w, 3) for w in worst))) +
          f" | per single view: mean {sum(per_view) / 8:.3f} max {max(per_view):.3f}")
