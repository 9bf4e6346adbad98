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
