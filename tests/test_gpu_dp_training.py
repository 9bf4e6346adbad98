"""Data-parallel training step, rehearsed with TWO ranks sharing the one GPU of the test box over gloo (RCCL refuses
two ranks on one device; the collectives are the same torch.distributed calls): every rank renders and
back-propagates its half of the ray batch, GradExchange sums the gradients -- they must equal the single-process
gradients of the whole batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N, SR, K, P = 60000, 32, 8, 12


def _setup(device):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import camera_rays, small_scene
    from pointnerf2studio_amd import synthetic
    from pointnerf2studio_amd.renderer import RendererHIP, SceneHIP, WeightsHIP, grid_hyperparameters
    pts = small_scene(N)
    w = synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1)
    xyz = pts["xyz"].to(device)
    hyp = grid_hyperparameters(xyz, (0.004, 0.004, 0.004), (2, 2, 2), (3, 3, 3), list(synthetic.CHAIR_RANGES))
    scene = SceneHIP()
    scene.build(xyz, hyp.ranges, hyp.scaled_vsize, hyp.scaled_vdim, (3, 3, 3), (3, 3, 3), P, 410000, True)
    scene.pack_points(xyz, pts["embedding"].to(device), pts["conf"].to(device), pts["dir"].to(device),
                      pts["color"].to(device))
    wh = WeightsHIP()
    wh.pack(w, pts["Rw2c"], device)
    rnd = RendererHIP(scene, wh, SR=SR, K=K, precision="fp32", eval_clamp=False)
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    g = torch.Generator().manual_seed(4)
    G = torch.randn(dirs.shape[0], 3, generator=g)
    return rnd, w, campos, camrot, dirs.to(device), G.to(device)


def _step(rnd, w, campos, camrot, dirs, G):
    rnd.render(dirs, campos, camrot, 2.0, 6.0)
    g = rnd.backward(G, w, N)
    return g, rnd.touched_points()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnerf2studio_amd.distributed import GradExchange
        from pointnerf2studio_amd.renderer import MLP_TENSOR_ORDER
        dev = torch.device("cuda:0")
        rnd, w, campos, camrot, dirs, G = _setup(dev)
        mine = torch.arange(rank, dirs.shape[0], world, device=dev)     # interleaved ray shard
        g, touched = _step(rnd, w, campos, camrot, dirs[mine].contiguous(), G[mine].contiguous())
        ex = GradExchange(average=False)
        # gloo moves host memory: stage through the CPU (on RCCL the same calls take the device tensors)
        e, c, d = g["embedding"].cpu(), g["color"].cpu(), g["dir"].cpu()
        ex.reduce_points(touched.cpu(), e, c, d)
        mlp = [g[n + s].cpu() for n in MLP_TENSOR_ORDER for s in (".weight", ".bias")]
        ex.reduce_mlp(mlp)
        if rank == 0:
            # numpy: pickled by value (a torch tensor would travel as a file descriptor of this process)
            q.put({"embedding": e.numpy(), "color": c.numpy(), "dir": d.numpy(),
                   **{n + s: mlp[2 * i + j].numpy() for i, n in enumerate(MLP_TENSOR_ORDER)
                      for j, s in enumerate((".weight", ".bias"))}})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_training_step_equals_single_process(gpu_device):
    rnd, w, campos, camrot, dirs, G = _setup(gpu_device)
    whole, touched = _step(rnd, w, campos, camrot, dirs, G)
    assert touched.numel() > 100 and int((whole["embedding"].abs().sum(1) > 0).sum()) <= touched.numel()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k, v in got.items():
        ref = whole[k].cpu()
        scale = ref.abs().max().item()
        assert (torch.from_numpy(v) - ref).abs().max().item() <= 2e-3 * scale, k
