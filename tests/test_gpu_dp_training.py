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


# ---- the same step through the PLUGIN: PointNerf under wrap_data_parallel (studio_pipeline.py:48-53) ----------------
def _plugin_model(device):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from helpers import camera_rays, small_scene
    from pointnerf2studio_amd import synthetic
    from pointnerf2studio_amd.model import PointNerf, PointNerfConfig
    from pointnerf2studio_amd.ns_compat import RayBundle
    pts = small_scene(N)
    sd = {"neural_points.xyz": pts["xyz"], "neural_points.points_embeding": pts["embedding"],
          "neural_points.points_conf": pts["conf"], "neural_points.points_dir": pts["dir"],
          "neural_points.points_color": pts["color"], "neural_points.Rw2c": pts["Rw2c"]}
    cfg = PointNerfConfig(ranges=list(synthetic.CHAIR_RANGES), max_o=410000, SR=SR, K=K, P=P, enable_collider=False)
    model = PointNerf(cfg, point_state_dict=sd).to(device)
    model.load_state_dict(synthetic.make_weights(0, sigma_scale=300.0, bias_scale=0.1), strict=False)
    model.train()
    model.neural_points.jitter = 0.0
    campos, camrot, dirs = camera_rays(32, 32, az=35.0)
    G = torch.randn(dirs.shape[0], 3, generator=torch.Generator().manual_seed(4)).to(device)

    def bundle(sel):
        n = sel.numel()
        return RayBundle(origins=campos[None].expand(n, 3).to(device), directions=dirs.to(device)[sel].contiguous(),
                         nears=torch.full((n, 1), 2.0, device=device), fars=torch.full((n, 1), 6.0, device=device),
                         metadata={"camrotc2w": camrot.to(device)})
    return model, bundle, G, dirs.shape[0]


def _plugin_loss(model_or_ddp, model, bundle, G, sel):
    out = model_or_ddp(bundle(sel))
    # a sum over rays (per-rank losses add up to the whole batch's) + the conf regulariser's own term, also as a sum
    # (the mean over the rank's slots times their number)
    conf = out["conf_coefficient_loss_term"] * out["conf_coefficient_slots"] * 1e-6
    return (out["coarse_raycolor"] * G[sel]).sum() + conf


def _plugin_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnerf2studio_amd.distributed import POINT_TENSORS_EXCHANGED_SPARSELY, wrap_data_parallel
        dev = torch.device("cuda:0")
        model, bundle, G, R = _plugin_model(dev)
        ddp = wrap_data_parallel(model, average=False)          # sums, to compare with the single-process gradients
        assert model.grad_exchange is not None and model.grad_exchange.world == world
        sel = torch.arange(rank, R, world, device=dev)
        _plugin_loss(ddp, model, bundle, G, sel).backward()
        grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
        # DDP averaged what it reduced (MLP, points_conf): undo, the exchange summed the point rows
        out = {n: (g if n in POINT_TENSORS_EXCHANGED_SPARSELY else g * world).cpu().numpy() for n, g in grads.items()}
        if rank == 0:
            q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_plugin_step_equals_single_process(gpu_device):
    """PointNerf under wrap_data_parallel, two ranks on one GPU over gloo: DDP all-reduces the MLP (and points_conf),
    the three big point tensors are kept out of DDP and their touched rows exchanged inside the fused backward.  The
    result equals one process differentiating the whole batch."""
    model, bundle, G, R = _plugin_model(gpu_device)
    _plugin_loss(model, model, bundle, G, torch.arange(R, device=gpu_device)).backward()
    whole = {n: p.grad.clone().cpu() for n, p in model.named_parameters() if p.grad is not None}
    assert {"neural_points.points_embeding", "neural_points.points_conf", "mlp_base.layers.0.weight"} <= set(whole)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_plugin_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert set(got) == set(whole)
    for k, v in got.items():
        ref = whole[k]
        scale = ref.abs().max().item()
        assert (torch.from_numpy(v) - ref).abs().max().item() <= 2e-3 * scale + 1e-12, k


# ---- several optimiser steps, data-parallel, with the optimisers studio_config registers --------------------------------
DP_STEPS = 6


def _dp_train(model_or_ddp, model, bundle, G, sel, R_total, world):
    """DP_STEPS steps of torch Adam (MLPs) + PointRowAdam (points) on the rank's rays `sel`; the loss is this rank's share of
    a mean over ALL rays, so that DDP's average over ranks is the whole batch's gradient."""
    from pointnerf2studio_amd.optim import PointRowAdam
    groups = model.get_param_groups()
    opt_f = torch.optim.Adam(groups["fields"], lr=5e-4, eps=1e-8)
    opt_p = PointRowAdam(groups["neural_points"], lr=2e-3, eps=1e-8)
    callbacks = model.get_training_callbacks(None)
    losses = []
    for it in range(DP_STEPS):
        opt_f.zero_grad(set_to_none=True)
        opt_p.zero_grad(set_to_none=True)
        out = model_or_ddp(bundle(sel))
        conf = out["conf_coefficient_loss_term"] * out["conf_coefficient_slots"] * 1e-6
        loss = ((out["coarse_raycolor"] * G[sel]).sum() + conf) * (world / R_total)
        loss.backward()
        opt_f.step()
        opt_p.step()
        for cb in callbacks:
            cb.run_callback(step=it)
        losses.append(float(loss.detach()) / world)
    assert opt_p.dense_steps == 0
    return losses, opt_p


def _dp_steps_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnerf2studio_amd.distributed import wrap_data_parallel
        dev = torch.device("cuda:0")
        model, bundle, G, R = _plugin_model(dev)
        ddp = wrap_data_parallel(model)                       # DDP's own semantics: the average over ranks
        sel = torch.arange(rank, R, world, device=dev)
        losses, opt_p = _dp_train(ddp, model, bundle, G, sel, R, world)
        params = {n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}
        # every rank must hold the same bits after every step: compare a checksum of all parameters across the ranks
        flat = torch.cat([p.reshape(-1).double() for p in params.values()])
        mine = torch.stack([flat.sum(), (flat * flat).sum()])
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        same = all(torch.equal(b, both[0]) for b in both)
        total = torch.tensor([sum(losses)], dtype=torch.float64)
        dist.all_reduce(total)                                # the ranks' loss shares add up to the batch's loss
        if rank == 0:
            q.put({"params": {n: p.numpy() for n, p in params.items()}, "same_on_all_ranks": same,
                   "loss_sum": float(total), "ever": opt_p.ever_touched()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_training_steps_with_the_registered_optimisers(gpu_device):
    """Six optimiser steps under wrap_data_parallel on two ranks (torch Adam for the MLPs through DDP's all-reduce,
    PointRowAdam for the point tensors over the UNION of the ranks' rows -- what the row exchange returns and publishes):
    the ranks stay bit-identical to each other, and end where one process training on the whole batch ends."""
    model, bundle, G, R = _plugin_model(gpu_device)
    losses, opt_p = _dp_train(model, model, bundle, G, torch.arange(R, device=gpu_device), R, 1)
    whole = {n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_dp_steps_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=420)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got["same_on_all_ranks"], "the ranks' parameters drifted apart"
    assert abs(got["loss_sum"] - sum(losses)) <= 1e-4 * abs(sum(losses)), (got["loss_sum"], sum(losses))
    assert got["ever"] == opt_p.ever_touched()                # the union of the ranks' rows = the whole batch's rows
    assert set(got["params"]) == set(whole)
    for k, v in got["params"].items():
        ref = whole[k]
        rel = ((torch.from_numpy(v) - ref).double().norm() / ref.double().norm().clamp(min=1e-30)).item()
        assert rel <= 1e-4, f"{k}: relative L2 {rel:.3e} after {DP_STEPS} steps"
    moved = (whole["neural_points.points_embeding"] - _plugin_model(gpu_device)[0].neural_points.points_embeding.detach().cpu()).abs().max()
    assert moved.item() > 1e-3                                # (the steps did train something)
