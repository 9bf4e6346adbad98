"""SURVEY.md section 8e on the GPU: the ray-tile shard + ONE all_gather per step, rehearsed with THREE ranks sharing the
one GPU of the test box over gloo (RCCL refuses several ranks on one device; the collectives are the same
torch.distributed calls).  Every rank renders its 16x16-pixel tiles of two views in one pnr_render_camera_lists call (views =
pose + intrinsics, rays = the pixel ids of its shard, the tile owner rotated per view) in the lego-like configuration of BASELINE.json configs[2] at
reduced size, the ranks all_gather their tiles (distributed.gather_views), and every rank must hold the two full
images bit-identical to the single-process render of the whole frames -- RGB and depth -- at jitter 0 and at the
reference's coarse-sample jitter of 0.3 (the jitter stream of a ray from a camera is keyed on its view and pixel, not on
its position in a rank's call: a frame does not depend on the number of ranks or on the rotation of the tile owners)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

H, W, N_PTS, WORLD = 80, 96, 150_000, 3


def _setup(device, jitter=0.0):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from pointnerf2studio_amd import synthetic
    from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, View, WeightsHIP, grid_hyperparameters
    c = dict(synthetic.SCENE_CONFIGS["cfg2_lego_6m"])
    pts = synthetic.make_scene_points(c, N=N_PTS)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    xyz = pts["xyz"].to(device)
    hyp = grid_hyperparameters(xyz, [c["vsize"]] * 3, (2, 2, 2), (3, 3, 3), list(c["ranges"]))
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, (3, 3, 3), (3, 3, 3), c["P"], c["max_o"], True)
    scene.pack_points(xyz, pts["embedding"].to(device), pts["conf"].to(device), pts["dir"].to(device),
                      pts["color"].to(device))
    wh = WeightsHIP()
    wh.pack(w, pts["Rw2c"], device)
    rnd = RendererHIP(scene, wh, SR=c["SR"], K=c["K"], jitter=jitter, seed=11)
    views = []
    for v in (1, 6):
        campos, camrot = synthetic.make_scene_camera(c, v)
        views.append(View.from_angle(campos, camrot, H, W, 0.35, c["near"], c["far"]))   # narrow view: the object fills it
    return rnd, views


def _worker(rank, world, port, q, jitter):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnerf2studio_amd.distributed import gather_views, make_shard
        dev = torch.device("cuda:0")
        rnd, views = _setup(dev, jitter)
        # bench.py's shard: the tile owner rotates with the view's position in the step, every view has its own pixel
        # list (pnr_render_camera_lists)
        shard = make_shard(H, W, world, rank, rotate=True)
        out = rnd.render_camera(views, H, W, pixels=shard.view_pixels[:len(views)].to(torch.int32).to(dev))
        local = torch.cat([out["rgb"], out["depth"][:, None]], dim=1).cpu()     # gloo moves host memory
        images = gather_views(local, shard, len(views))                        # [2, H*W, 4] on every rank
        kept = int(out["counters"]["rays_kept"])
        q.put((rank, images.numpy(), kept))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("jitter", [0.0, 0.3])
def test_three_rank_tile_shard_equals_single_process_frames(gpu_device, jitter):
    rnd, views = _setup(gpu_device, jitter)
    whole = rnd.render_camera(views, H, W)
    want = torch.cat([whole["rgb"], whole["depth"][:, None]], dim=1).view(2, H * W, 4).cpu()
    assert whole["counters"]["rays_kept"] > 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q, jitter)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in got) == list(range(WORLD))
    for rank, images, kept in got:
        assert torch.equal(torch.from_numpy(images), want), f"rank {rank}: gathered frames differ from the single-process render"
    # the round-robin tile deal balances the hit rays: no rank holds more than 45 % of them (3 ranks)
    kepts = [k for _, _, k in got]
    assert max(kepts) <= 0.45 * sum(kepts), kepts
