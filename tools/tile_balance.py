"""Diagnostic: load balance of the round-robin tile shard (pairs per rank and per view) for several tile sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointnerf2studio_amd import synthetic
from pointnerf2studio_amd.distributed import make_shard
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
H = W = 800
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for tile in (32, 16, 8, 4):
    worst, tot = [], []
    for az in [20.0, 65.0, 110.0, 155.0, 200.0, 245.0, 290.0, 335.0]:
        campos, camrot = synthetic.make_camera(az)
        d = synthetic.make_rays(H, W, campos, camrot).to(dev)
        pairs, ms = [], []
        for r in range(world):
            sh = make_shard(H, W, world, r, tile=tile).to(dev)
            dirs = d.index_select(0, sh.pixels).contiguous()
            out = rnd.render(dirs, campos, camrot, 2.0, 6.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rnd.render(dirs, campos, camrot, 2.0, 6.0, sync_counters=False); e1.record(); torch.cuda.synchronize()
            pairs.append(out["counters"]["pairs_valid"]); ms.append(e0.elapsed_time(e1))
        worst.append(max(ms) / (sum(ms) / world)); tot.append(sum(ms))
    print(f"tile {tile:2d}: max/mean render time over ranks, per view: " + " ".join(f"{x:.3f}" for x in worst) +
          f" | mean {sum(worst)/len(worst):.3f} | sum of per-rank ms over all views {sum(tot):.1f}")
