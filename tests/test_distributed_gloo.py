"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  The render itself is HIP-only, so the per-rank
"render" here is a deterministic function of the pixel id; what is tested is everything that surrounds it in
bench.py's step: tile ownership, padding, all_gather_into_tensor, de-interleave, max-over-ranks timing."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointnerf2studio_amd.distributed import ViewGatherPipe, gather_image, gather_views, make_shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(pixels):
    p = pixels.to(torch.float32)
    return torch.stack([p, p * 0.5, -p, p % 7], dim=1)


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = make_shard(H, W, world, rank)
        local = _fake_render(shard.pixels)
        img = gather_image(local, shard)
        expect = _fake_render(torch.arange(H * W))
        ok = torch.equal(img, expect)
        # bench.py's step: `world` views rendered in one call, one all_gather for all of them
        V = world
        local_v = torch.cat([_fake_render(shard.pixels) + 1000.0 * v for v in range(V)])
        imgs = gather_views(local_v, shard, V)
        ok = ok and all(torch.equal(imgs[v], expect + 1000.0 * v) for v in range(V))
        # ... and the pipelined form bench.py uses (the collective of step s overlaps step s + 1): every step's
        # images must equal the synchronous result, including the last ones returned by drain()
        pipe = ViewGatherPipe(shard, V, 4, torch.float32, torch.device("cpu"))
        seen = []
        for step in range(4):
            pipe.stage().copy_(local_v + 7.0 * step)
            pipe.submit()
            if step > 0:
                seen.append(pipe.images.clone())      # images of step - 1 were assembled by this submit
        seen.append(pipe.drain().clone())
        ok = ok and len(seen) == 4 and all(
            torch.equal(seen[st][v], expect + 1000.0 * v + 7.0 * st) for st in range(4) for v in range(V))
        # bench.py's shard: the tile owner rotates with the view's position in the step (view i of rank q holds the
        # tiles of owner (q + i) % world): same images, synchronous and pipelined
        rsh = make_shard(H, W, world, rank, rotate=True)
        ok = ok and rsh.rotate and rsh.view_pixels.shape == (world, rsh.n_pad) and torch.equal(rsh.view_pixels[0], rsh.pixels)
        local_r = torch.cat([_fake_render(rsh.pixels_of_view(v)) + 1000.0 * v for v in range(V)])
        imgs_r = gather_views(local_r, rsh, V)
        ok = ok and all(torch.equal(imgs_r[v], expect + 1000.0 * v) for v in range(V))
        pipe_r = ViewGatherPipe(rsh, V, 4, torch.float32, torch.device("cpu"))
        for step in range(3):
            pipe_r.stage().copy_(local_r + 7.0 * step)
            pipe_r.submit()
        last = pipe_r.drain()
        ok = ok and all(torch.equal(last[v], expect + 1000.0 * v + 14.0) for v in range(V))
        # bench.py's timing reduction: max over ranks
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok, float(t.item()), shard.n_valid, shard.n_pad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 64, 48), (2, 50, 50), (3, 40, 56)])
def test_tile_shard_all_gather(world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _, _ in res)
    assert all(t == float(world) for _, _, t, _, _ in res)
    assert sum(nv for _, _, _, nv, _ in res) == H * W
    assert len({npad for _, _, _, _, npad in res}) == 1


def test_shards_partition_the_image_and_interleave():
    for H, W, world in [(800, 800, 8), (800, 800, 4), (64, 64, 2), (1296, 968, 8)]:
        shards = [make_shard(H, W, world, r) for r in range(world)]
        allp = torch.cat([s.pixels[:s.n_valid] for s in shards])
        assert allp.numel() == H * W and torch.equal(torch.sort(allp)[0], torch.arange(H * W))
        counts = [s.n_valid for s in shards]
        assert max(counts) - min(counts) <= 2 * 16 * 16   # balanced to within a couple of tiles
        # interleaving: every rank owns tiles in the top AND the bottom half of the image
        for s in shards:
            ys = s.pixels[:s.n_valid] // W
            assert ys.min() < H // 4 and ys.max() > 3 * H // 4
        # rotated owners: for every view position the ranks' lists still partition the image
        rot = [make_shard(H, W, world, r, rotate=True) for r in range(world)]
        for i in range(world):
            owners = [(r + i) % world for r in range(world)]
            allp = torch.cat([rot[r].pixels_of_view(i)[:shards[owners[r]].n_valid] for r in range(world)])
            assert torch.equal(torch.sort(allp)[0], torch.arange(H * W))


# ---- data-parallel training step: gradient exchange ---------------------------------------------------------------
def _grad_worker(rank, world, port, N, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnerf2studio_amd.distributed import GradExchange

        def local_grads(r):
            g = torch.Generator().manual_seed(100 + r)
            u = 50 + 37 * r                                      # ranks touch different numbers of points ...
            idx = torch.randperm(N, generator=g)[:u]
            idx[0] = 3                                           # ... and share some of them
            idx[1] = N - 1 if N - 1 not in idx[2:].tolist() else idx[1]
            idx = idx.unique()
            emb, col, dr = torch.zeros(N, 32), torch.zeros(N, 3), torch.zeros(N, 3)
            emb[idx] = torch.randn(idx.numel(), 32, generator=g)
            col[idx] = torch.randn(idx.numel(), 3, generator=g)
            dr[idx] = torch.randn(idx.numel(), 3, generator=g)
            flat = torch.randn(1000, generator=g)
            return idx, emb, col, dr, flat

        idx, emb, col, dr, flat = local_grads(rank)
        everyone = [local_grads(r) for r in range(world)]
        ok = True
        for average in (False, True):
            e, c, d, f = emb.clone(), col.clone(), dr.clone(), flat.clone()
            ex = GradExchange(average=average)
            views = [f[:600].view(20, 30), f[600:]]              # views of one flat buffer: reduced without a copy
            ex.reduce_mlp(views)
            got_rows = ex.reduce_points(idx, e, c, d)
            scale = world if average else 1
            ok = ok and torch.allclose(e, sum(x[1] for x in everyone) / scale, atol=1e-6)
            ok = ok and torch.allclose(c, sum(x[2] for x in everyone) / scale, atol=1e-6)
            ok = ok and torch.allclose(d, sum(x[3] for x in everyone) / scale, atol=1e-6)
            ok = ok and torch.allclose(f, sum(x[4] for x in everyone) / scale, atol=1e-6)
            ok = ok and got_rows == sum(x[0].numel() for r, x in enumerate(everyone) if r != rank)
            # separate tensors (not views of one buffer) take the copying path
            a, b = flat[:10].clone(), flat[10:30].clone()
            ex.reduce_mlp([a, b])
            ok = ok and torch.allclose(a, sum(x[4][:10] for x in everyone) / scale, atol=1e-6)
            ok = ok and torch.allclose(b, sum(x[4][10:30] for x in everyone) / scale, atol=1e-6)
        # the sparse form end to end (rows in, union rows out; no dense tensor): equals the dense sums on its rows,
        # covers exactly the points some rank touched, ascending
        for average in (False, True):
            sidx = torch.sort(idx)[0]
            rows = torch.cat([emb[sidx], col[sidx], dr[sidx], torch.zeros(sidx.numel(), 2)], dim=1)
            u_idx, u_rows = GradExchange(average=average).reduce_points_sparse(sidx, rows)
            scale = world if average else 1
            want_idx = torch.cat([x[0] for x in everyone]).unique()
            ok = ok and torch.equal(u_idx, want_idx)
            ok = ok and torch.allclose(u_rows[:, :32], sum(x[1] for x in everyone)[want_idx] / scale, atol=1e-6)
            ok = ok and torch.allclose(u_rows[:, 32:35], sum(x[2] for x in everyone)[want_idx] / scale, atol=1e-6)
            ok = ok and torch.allclose(u_rows[:, 35:38], sum(x[3] for x in everyone)[want_idx] / scale, atol=1e-6)
            ok = ok and float(u_rows[:, 38:].abs().sum()) == 0.0
        # a rank whose rays hit nothing contributes zero rows
        e, c, d = torch.zeros(N, 32), torch.zeros(N, 3), torch.zeros(N, 3)
        mine = idx if rank == 0 else idx[:0]
        if rank == 0:
            e[idx] = 1.0
        GradExchange(average=False).reduce_points(mine, e, c, d)
        ok = ok and float(e.sum()) == 32.0 * everyone[0][0].numel()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 5000), (3, 20_000_000 // 1000)])
def test_gradient_exchange_sparse_points_and_flat_mlp(world, N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_gradient_exchange_is_a_no_op_on_one_rank():
    from pointnerf2studio_amd.distributed import GradExchange
    ex = GradExchange(world=1)
    e = torch.ones(10, 32)
    assert ex.reduce_points(torch.arange(3), e, torch.zeros(10, 3), torch.zeros(10, 3)) == 0
    t = torch.ones(5)
    ex.reduce_mlp([t])
    assert torch.equal(e, torch.ones(10, 32)) and torch.equal(t, torch.ones(5))


def test_point_index_survives_the_float_block():
    from pointnerf2studio_amd.distributed import _index_from_f32_halves, _index_to_f32_halves
    idx = torch.tensor([0, 1, 16777215, 16777216, 16777217, 19_999_999, 2 ** 31 + 5, 2 ** 40 + 123456789])
    lo, hi = _index_to_f32_halves(idx)
    assert torch.equal(_index_from_f32_halves(lo, hi), idx)
