"""Multi-GPU ray-tile sharding: one process per GPU, point cloud + voxel structure + weights replicated,
the image cut into 16x16-pixel tiles dealt round-robin to the ranks, ONE all_gather of the rendered tiles
per image (RCCL over xGMI via torch.distributed backend "nccl"; "gloo" in the CPU tests).

The reference has no notion of this (its only collective is DDP's gradient all-reduce,
studio_pipeline.py:48-53); rays are independent, so the render path shards with a single exchange step.
Round-robin tiles rather than contiguous slabs: object pixels cluster, and per-ray cost varies by >10x
between background and surface rays.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


@dataclass
class TileShard:
    H: int
    W: int
    tile: int
    world: int
    rank: int
    pixels: torch.Tensor       # [n_pad] flat pixel ids (row-major) rendered by this rank, padded
    n_valid: int               # the first n_valid entries of `pixels` are real
    n_pad: int                 # identical on every rank (all_gather needs equal sizes)
    scatter_index: torch.Tensor  # [world * n_pad] flat pixel id of every gathered slot, -1 for padding
    slot_index: torch.Tensor   # [H*W] gathered slots that hold real pixels ...
    pixel_index: torch.Tensor  # [H*W] ... and the pixel each of them is
    # rotate=True (multi-view steps): view i of a step is rendered with the tiles of owner (rank + i) % world, so a
    # rank's load is a sum over DIFFERENT tile sets -- the views of a step look at the same object, and with one tile
    # set for all of them a rank's imbalance repeats in every view (max / mean pairs per rank 1.05 at 8 ranks; 1.004
    # rotated).  view_pixels [world, n_pad]: row i = the list this rank renders for the step's i-th view.
    view_pixels: Optional[torch.Tensor] = None

    @property
    def rotate(self) -> bool:
        return self.view_pixels is not None

    def pixels_of_view(self, i: int) -> torch.Tensor:
        return self.view_pixels[i % self.world] if self.rotate else self.pixels

    def to(self, device) -> "TileShard":
        return TileShard(self.H, self.W, self.tile, self.world, self.rank, self.pixels.to(device), self.n_valid,
                         self.n_pad, self.scatter_index.to(device), self.slot_index.to(device),
                         self.pixel_index.to(device),
                         None if self.view_pixels is None else self.view_pixels.to(device))


def _rank_pixels(H: int, W: int, tile: int, world: int, rank: int) -> torch.Tensor:
    nty, ntx = (H + tile - 1) // tile, (W + tile - 1) // tile
    ids = torch.arange(nty * ntx)
    mine = ids[ids % world == rank]
    ty, tx = mine // ntx, mine % ntx
    yy = (ty[:, None, None] * tile + torch.arange(tile)[None, :, None]).expand(-1, tile, tile)
    xx = (tx[:, None, None] * tile + torch.arange(tile)[None, None, :]).expand(-1, tile, tile)
    ok = (yy < H) & (xx < W)
    return (yy * W + xx)[ok].reshape(-1)


def make_shard(H: int, W: int, world: int, rank: int, tile: int = 16, rotate: bool = False) -> TileShard:
    """Pixel ownership of `rank`; deterministic and identical on all ranks (no communication).  rotate: see
    TileShard.view_pixels."""
    per_rank = [_rank_pixels(H, W, tile, world, r) for r in range(world)]
    n_pad = max(int(p.numel()) for p in per_rank)
    scatter = torch.full((world * n_pad,), -1, dtype=torch.long)
    for r, p in enumerate(per_rank):
        scatter[r * n_pad: r * n_pad + p.numel()] = p
    mine = per_rank[rank]
    n_valid = int(mine.numel())
    if n_valid < n_pad:  # pad with the rank's last pixel: rendered twice, dropped by scatter_index
        mine = torch.cat([mine, mine[-1:].expand(n_pad - n_valid)])
    slot_index = torch.nonzero(scatter >= 0).reshape(-1)
    view_pixels = None
    if rotate and world > 1:
        def padded(p):   # every owner's list padded with its last pixel to the common length
            return p if p.numel() == n_pad else torch.cat([p, p[-1:].expand(n_pad - p.numel())])
        view_pixels = torch.stack([padded(per_rank[(rank + i) % world]) for i in range(world)]).contiguous()
    return TileShard(H, W, tile, world, rank, mine.contiguous(), n_valid, n_pad, scatter, slot_index,
                     scatter[slot_index].contiguous(), view_pixels)


def _owner_major(gathered: torch.Tensor, shard: TileShard) -> torch.Tensor:
    """gathered [world (rank q), V, n_pad, C] -> [V, world (tile owner o), n_pad, C].  Without rotation rank = owner;
    with it view i of rank q holds owner (q + i) % world, so owner o of view i comes from rank (o - i) % world."""
    g = gathered.permute(1, 0, 2, 3)
    if not shard.rotate:
        return g
    V, w = g.shape[0], shard.world
    i = torch.arange(V, device=g.device)[:, None]
    o = torch.arange(w, device=g.device)[None, :]
    return g[i, (o - i) % w]


def gather_views(local: torch.Tensor, shard: TileShard, n_views: int, out: Optional[torch.Tensor] = None,
                 gathered: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Several views rendered in one call: local [n_views * n_pad, C] (view-major, each view in shard.pixels order)
    -> images [n_views, H*W, C] on every rank with ONE all_gather_into_tensor + one index_copy."""
    C = local.shape[1]
    w, n = shard.world, shard.n_pad
    if gathered is None:
        gathered = torch.empty((w, n_views, n, C), dtype=local.dtype, device=local.device)
    if w > 1:
        dist.all_gather_into_tensor(gathered.view(-1, C), local.contiguous())
    else:
        gathered.view(-1, C).copy_(local)
    if out is None:
        out = torch.empty((n_views, shard.H * shard.W, C), dtype=local.dtype, device=local.device)
    # [world, V, n_pad, C] -> [V, world * n_pad, C]: the per-view layout gather_image produces
    per_view = _owner_major(gathered, shard).reshape(n_views, w * n, C)
    src = per_view if shard.slot_index.numel() == w * n else per_view.index_select(1, shard.slot_index)
    out.index_copy_(1, shard.pixel_index, src)
    return out


def gather_image(local: torch.Tensor, shard: TileShard, out: Optional[torch.Tensor] = None,
                 gathered: Optional[torch.Tensor] = None) -> torch.Tensor:
    """local [n_pad, C] (this rank's rendered pixels in shard.pixels order) -> full image [H*W, C] on EVERY
    rank: one all_gather_into_tensor + one index_copy.  `shard` must live on local's device (TileShard.to);
    `gathered` / `out` may be passed in pre-allocated to keep the step allocation-free."""
    C = local.shape[1]
    if gathered is None:
        gathered = torch.empty((shard.world * shard.n_pad, C), dtype=local.dtype, device=local.device)
    if shard.world > 1:
        dist.all_gather_into_tensor(gathered, local.contiguous())
    else:
        gathered.copy_(local)
    if out is None:
        out = torch.empty((shard.H * shard.W, C), dtype=local.dtype, device=local.device)
    src = gathered if shard.slot_index.numel() == gathered.shape[0] else gathered.index_select(0, shard.slot_index)
    out.index_copy_(0, shard.pixel_index, src)
    return out


class ViewGatherPipe:
    """bench.py's exchange step, overlapped with the next step's render.

    Per step a rank fills `stage()` ([n_views * n_pad, C], view-major, each view in shard.pixels order) and calls
    `submit()`: the all_gather of that step starts on the backend's own stream (async_op) while the caller goes on
    to render the next step; the images of the PREVIOUS step are assembled (one index_copy) at that point, the last
    ones by `drain()`.  Two sets of buffers alternate, so a buffer is rewritten only after its collective has been
    waited for.  `local_copy=True` (single rank, or the one-GPU emulation of an N-rank step) replaces the collective
    by a copy of the rank's own slice."""

    def __init__(self, shard: TileShard, n_views: int, C: int, dtype, device, local_copy: bool = False):
        w, n = shard.world, shard.n_pad
        self.shard, self.n_views, self.C = shard, n_views, C
        self.use_dist = w > 1 and not local_copy
        self.local = [torch.empty((n_views * n, C), dtype=dtype, device=device) for _ in range(2)]
        self.gathered = [torch.empty((w, n_views, n, C), dtype=dtype, device=device) for _ in range(2)]
        self.images = torch.empty((n_views, shard.H * shard.W, C), dtype=dtype, device=device)
        self.work = [None, None]
        self.use_async = True
        self.step = 0
        self.prev = None

    def stage(self) -> torch.Tensor:
        return self.local[self.step % 2]

    def submit(self) -> None:
        b = self.step % 2
        if self.use_dist:
            out, inp = self.gathered[b].view(-1, self.C), self.local[b]
            if self.use_async:
                try:
                    self.work[b] = dist.all_gather_into_tensor(out, inp, async_op=True)
                except (RuntimeError, NotImplementedError):
                    self.use_async = False   # a backend without asynchronous all_gather: exchange synchronously
            if not self.use_async:
                dist.all_gather_into_tensor(out, inp)
        else:
            self.gathered[b][0].view(-1, self.C).copy_(self.local[b])
        if self.prev is not None:
            self._assemble(self.prev)
        self.prev = b
        self.step += 1

    def drain(self) -> torch.Tensor:
        if self.prev is not None:
            self._assemble(self.prev)
            self.prev = None
        return self.images

    def _assemble(self, b: int) -> None:
        if self.work[b] is not None:
            self.work[b].wait()   # NCCL/RCCL: the current stream waits for the collective; gloo: the host does
            self.work[b] = None
        sh, w, n = self.shard, self.shard.world, self.shard.n_pad
        per_view = _owner_major(self.gathered[b], sh).reshape(self.n_views, w * n, self.C)
        src = per_view if sh.slot_index.numel() == w * n else per_view.index_select(1, sh.slot_index)
        self.images.index_copy_(1, sh.pixel_index, src)


# ---------------------------------------------------------------------------------------------------------------
# Data-parallel training step: every rank renders + back-propagates its share of the ray batch against the replicated
# cloud; the gradients meet in two collectives.  (The reference's only collective is DDP's dense all-reduce over
# every parameter, studio_pipeline.py:48-53 -- 768 MB of `points_embeding.grad` per step at 6 M points, of which a
# 4096-ray batch touches ~60 k rows.)
# ---------------------------------------------------------------------------------------------------------------
def _index_to_f32_halves(idx: torch.Tensor):
    """Point indices as two fp32 columns (low 24 bits, the rest): exact for indices below 2^48."""
    return (idx % 16777216).to(torch.float32), (idx // 16777216).to(torch.float32)


def _index_from_f32_halves(lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    return lo.to(torch.long) + hi.to(torch.long) * 16777216


POINT_TENSORS_EXCHANGED_SPARSELY = ("neural_points.points_embeding", "neural_points.points_color",
                                    "neural_points.points_dir")


def wrap_data_parallel(model, device_ids=None, average: bool = True, **ddp_kwargs):
    """The DDP wrap of studio_pipeline.py:48-53 for the fused training step.  The reference hands EVERY parameter to
    DDP: 768 MB of `points_embeding.grad` all-reduced per step at 6 M points, of which a 4096-ray batch touches ~60 k
    rows.  Here the three point tensors whose gradients the fused backward writes itself are excluded from DDP
    (`_set_params_and_buffers_to_ignore_for_model`) and exchanged as ROWS inside the backward
    (`GradExchange.exchange_dense_rows`, ~10 MB per rank); DDP keeps all-reducing the rest: the nine Linear layers
    (1.4 MB, one bucket) and `points_conf` (its gradient comes from torch autograd, dense [1,N,1]).  Needs
    hip_fused_training and hip_sparse_point_grads (the defaults); otherwise the plain wrap of the reference is returned.
    Returns the DDP module; `model.grad_exchange` is set."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    cfg = getattr(model, "config", None)
    fused = (getattr(cfg, "hip_fused_training", True) and getattr(cfg, "hip_sparse_point_grads", True)
             and model._fusable())
    ddp_kwargs.setdefault("find_unused_parameters", True)
    if fused:
        DDP._set_params_and_buffers_to_ignore_for_model(model, list(POINT_TENSORS_EXCHANGED_SPARSELY))
        model.grad_exchange = GradExchange(average=average)
    return DDP(model, device_ids=device_ids, **ddp_kwargs)


class GradExchange:
    """MLP gradients: ONE all_reduce of a flat 1.4-MB buffer.  Point gradients: SPARSE -- only the rows of the
    neighbour points this rank's rays touched travel: one all_gather of the row counts, one of the padded
    [U_max, 40] (index | d_embedding | d_color | d_dir) blocks, then an index_add of the other ranks' rows into
    the local dense gradients.  `average` divides by the world size (DDP semantics: every rank's loss is a mean over
    its own rays)."""

    def __init__(self, world: Optional[int] = None, average: bool = True):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.average = average

    @staticmethod
    def _all_gather(out: torch.Tensor, inp: torch.Tensor) -> None:
        """all_gather_into_tensor; staged through host memory on a backend that moves host memory only (gloo: the CPU
        rehearsals of this module, and two ranks sharing one GPU in tests/test_gpu_dp_training.py)."""
        if inp.is_cuda and dist.get_backend() == "gloo":
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp)

    def reduce_mlp(self, tensors: List[torch.Tensor]) -> None:
        """In place.  `tensors` are the weight / bias gradients; views of one flat buffer are reduced without a copy."""
        if self.world == 1 or not tensors:
            return
        base = tensors[0]._base if tensors[0]._base is not None else None
        same = base is not None and all(t._base is base for t in tensors) and base.dim() == 1
        flat = base if same else torch.cat([t.reshape(-1) for t in tensors])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if self.average:
            flat.div_(self.world)
        if not same:
            off = 0
            for t in tensors:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    def reduce_points_sparse(self, point_index: torch.Tensor, point_grads: torch.Tensor):
        """The sparse form end to end: this rank's rows as pnr_render_backward emits them (`point_index` [U] int64
        ascending, `point_grads` [U, 40] = [d_embedding 32 | d_color 3 | d_dir 3 | 0 0]; RendererHIP.backward(...,
        sparse_points=True)) in, the union over all ranks out: (index [U_all] ascending, rows [U_all, 40]), rows of a
        point several ranks touched summed in rank order (one index_add_ per rank: bitwise repeatable on a GPU too).  No
        dense [N, .] tensor exists on the way:
        the optimiser (or an index_add_ into .grad) consumes the rows."""
        dev = point_grads.device
        idx = point_index.to(device=dev, dtype=torch.long).reshape(-1)
        if self.world == 1:
            return idx, point_grads
        u = torch.tensor([idx.numel()], dtype=torch.long, device=dev)
        counts = torch.empty(self.world, dtype=torch.long, device=dev)
        self._all_gather(counts, u)
        counts_h = counts.tolist()
        u_max = max(max(counts_h), 1)
        block = torch.zeros((u_max, 40), dtype=torch.float32, device=dev)
        n = idx.numel()
        block[:n, :38] = point_grads[:, :38]
        block[:n, 38], block[:n, 39] = _index_to_f32_halves(idx)     # the two pad columns carry the index
        gathered = torch.empty((self.world * u_max, 40), dtype=torch.float32, device=dev)
        self._all_gather(gathered, block)
        per_rank = [gathered[r * u_max: r * u_max + counts_h[r]] for r in range(self.world)]
        all_idx = torch.cat([_index_from_f32_halves(rows[:, 38], rows[:, 39]) for rows in per_rank])
        uniq, inv = torch.unique(all_idx, sorted=True, return_inverse=True)
        out = torch.zeros((uniq.numel(), 40), dtype=torch.float32, device=dev)
        # rank order inside every point: ONE index_add_ per rank, in rank order.  A point appears at most once in a
        # rank's rows, so no two adds of one call meet in a row (on a GPU index_add_ is made of float atomics: a single
        # call over all ranks' rows would leave the order of a point's adds to the scheduler)
        off = 0
        for rows in per_rank:
            n_r = rows.shape[0]
            if n_r:
                out[:, :38].index_add_(0, inv[off:off + n_r], rows[:, :38])
            off += n_r
        if self.average:
            out.div_(self.world)
        return uniq, out

    def reduce_points(self, touched: torch.Tensor, d_embedding: torch.Tensor, d_color: torch.Tensor,
                      d_dir: torch.Tensor) -> int:
        """In place on the dense [N,32] / [N,3] / [N,3] gradients.  `touched`: the (unique) point indices this rank's
        gradient rows live in (int64).  Returns the total number of rows received from other ranks."""
        if self.world == 1:
            return 0
        dev = d_embedding.device
        touched = touched.to(device=dev, dtype=torch.long).reshape(-1)
        u = torch.tensor([touched.numel()], dtype=torch.long, device=dev)
        counts = torch.empty(self.world, dtype=torch.long, device=dev)
        self._all_gather(counts, u)
        counts_h = counts.tolist()
        u_max = max(max(counts_h), 1)
        # one fixed-width block per rank: [index low 24 bits | index high bits | d_embedding 32 | d_color 3 | d_dir 3]
        # (the index halves are exact in fp32; one dtype keeps it to ONE collective)
        block = torch.zeros((u_max, 40), dtype=torch.float32, device=dev)
        n = touched.numel()
        block[:n, 0], block[:n, 1] = _index_to_f32_halves(touched)
        block[:n, 2:34] = d_embedding[touched]
        block[:n, 34:37] = d_color[touched]
        block[:n, 37:40] = d_dir[touched]
        gathered = torch.empty((self.world * u_max, 40), dtype=torch.float32, device=dev)
        self._all_gather(gathered, block)
        me = dist.get_rank()
        received = 0
        seen = [touched]
        # the rows of a point are summed in RANK order on every rank (own rows taken out first, then every rank's rows
        # added back one index_add_ per rank: a point appears at most once per rank, so no call adds twice into a row):
        # all ranks end up with the same bits
        mine = (d_embedding[touched].clone(), d_color[touched].clone(), d_dir[touched].clone())
        d_embedding[touched] = 0
        d_color[touched] = 0
        d_dir[touched] = 0
        for r in range(self.world):
            if counts_h[r] == 0:
                continue
            if r == me:
                idx, e, c, d = touched, mine[0], mine[1], mine[2]
            else:
                rows = gathered[r * u_max: r * u_max + counts_h[r]]
                idx = _index_from_f32_halves(rows[:, 0], rows[:, 1])
                e, c, d = rows[:, 2:34], rows[:, 34:37], rows[:, 37:40]
                seen.append(idx)
                received += counts_h[r]
            d_embedding.index_add_(0, idx, e)
            d_color.index_add_(0, idx, c)
            d_dir.index_add_(0, idx, d)
        self.last_union = torch.cat(seen).unique()
        if self.average:
            # only rows some rank touched are non-zero: scale those
            all_idx = self.last_union
            d_embedding[all_idx] /= self.world
            d_color[all_idx] /= self.world
            d_dir[all_idx] /= self.world
        return received

    def exchange_dense_rows(self, index: torch.Tensor, count: torch.Tensor, targets):
        """The fused training step's point-gradient exchange (model._after_point_backward): this rank's backward has
        accumulated its rows into the dense tensors `targets` = {'embedding' [.., N, 32], 'color' [.., N, 3], 'dir'
        [.., N, 3]} (None = not trainable); `index` (int32, padded) / `count` (int64 [1], device) are the rows it touched
        (RendererHIP.touched).  Only those rows travel (reduce_points); returns the union of all ranks' rows in the same
        padded form, which is what the caller has to clean before the tensors are reused.  One host read (the count).
        The rows are sent as the tensors hold them: they must hold THIS step's contribution only.  Gradient accumulation
        over several backward calls and DDP's no_sync are not supported on this path -- PointNerf._point_grad_targets
        raises when a point tensor's .grad is already set before a data-parallel backward."""
        n = int(count.item())
        touched = index[:n].to(torch.long)
        ref = next(t for t in targets.values() if t is not None)
        N = ref.numel() // ref.shape[-1]
        dense = {}
        for key, width in (("embedding", 32), ("color", 3), ("dir", 3)):
            t = targets.get(key)
            dense[key] = t.view(N, width) if t is not None else torch.zeros((N, width), dtype=torch.float32,
                                                                            device=ref.device)
        self.reduce_points(touched, dense["embedding"], dense["color"], dense["dir"])
        union = self.last_union.to(torch.int32)
        return union, torch.tensor([union.numel()], dtype=torch.int64, device=union.device)
