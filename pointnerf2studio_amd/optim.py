"""`PointRowAdam`: torch.optim.Adam for the `neural_points` parameter group without the O(N) sweep.

The reference registers Adam at lr 2e-3 for the point tensors (studio_config.py:41-47).  torch's Adam updates all
N x 38 point values every step -- at 6 M points 4.2 ms of a 6 ms training step -- while a 4096-ray batch gives ~60 k
rows a gradient.  A row whose gradient has been zero since the optimiser was created has zero moments, and dense Adam
moves it by -step_size * 0 / (0 + eps) = 0: Adam over the rows that EVER had a gradient is the same optimiser.  The fused
training step knows which rows a backward wrote (`publish_rows`, called by model.PointNerf after every backward with the
device-side row list of pnr_render_touched); `step()` merges them into the ever-touched set (pnr_rows_merge) and applies
torch's update to that set for all point tensors in ONE launch (pnr_adam_rows).  Nothing is read back to the host.

State layout is torch.optim.Adam's (`step`, `exp_avg`, `exp_avg_sq`, dense tensors): checkpoints are interchangeable; after
`load_state_dict` the ever-touched set is recovered from the non-zero rows of exp_avg_sq (one O(N) pass).  A parameter
that received a gradient nobody published rows for (another backward path: torch autograd through index_select, DDP's
dense all-reduce) is updated densely on that step -- exact, merely not sparse -- and is dense from then on.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch.utils.weak import WeakTensorKeyDictionary

from . import _lib

# parameter -> the PointRowAdam that owns it (weakly, by identity: tensors compare elementwise, a plain WeakKeyDictionary
# cannot hold them).  publish_rows hands a backward's row list to the owning optimiser, which merges it into its
# ever-touched set RIGHT AWAY (one small launch, pnr_rows_merge): no list is kept, so nothing accumulates whether or not an
# optimiser step follows the backward (steps without an optimiser, gradient accumulation); under any other optimiser
# publish_rows is a no-op.  Rows of a backward whose gradient is later discarded only make the set a superset: a row with
# zero moments and a zero gradient moves by exactly 0.
_SUBSCRIBED = WeakTensorKeyDictionary()


def publish_rows(params, index: torch.Tensor, count: Optional[torch.Tensor]) -> None:
    """The rows (dim -2 of every tensor in `params`) a backward just wrote gradients into: `index` int32 on the device,
    of which the first min(count, len(index)) entries count (`count`: int64 [1] device tensor, or None = all)."""
    done = set()
    for p in params:
        if p is None:
            continue
        ref = _SUBSCRIBED.get(p)
        opt = ref() if ref is not None else None
        if opt is not None:
            opt._ingest(p, index, count, done)


def _rows_of(p: torch.Tensor) -> Tuple[int, int]:
    """(rows, width) of a point tensor in the reference's layouts: [1, N, C] or [N, C]."""
    if p.dim() < 2:
        return p.numel(), 1
    return p.numel() // p.shape[-1], p.shape[-1]


class _EverRows:
    """The set of rows that ever had a gradient, for parameters of one row count, on the device."""

    def __init__(self, num_rows: int, device):
        self.num_rows = num_rows
        self.flags = torch.zeros(num_rows, dtype=torch.int32, device=device)
        self.rows = torch.empty(num_rows, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        self.dense = False        # every row (a dense gradient arrived): no list needed any more

    def merge(self, lib, index: torch.Tensor, count: Optional[torch.Tensor], stream) -> None:
        idx = index.to(device=self.flags.device, dtype=torch.int32).contiguous()
        cnt = None if count is None else count.to(device=self.flags.device, dtype=torch.int64).contiguous()
        _lib.check(lib.pnr_rows_merge(self.flags.data_ptr(), self.num_rows, self.rows.data_ptr(), self.count.data_ptr(),
                                      self.num_rows, idx.data_ptr(), idx.numel(),
                                      None if cnt is None else cnt.data_ptr(), stream), "pnr_rows_merge")


class PointRowAdam(torch.optim.Optimizer):
    """Drop-in for torch.optim.Adam(params, lr, betas, eps) over point tensors whose gradients are row-sparse; same
    arguments as nerfstudio's AdamOptimizerConfig hands over (lr, eps, weight_decay = 0)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 always_rows=(0,)):
        if weight_decay != 0.0:
            raise ValueError("PointRowAdam: weight decay moves every row on every step; use torch.optim.Adam for that")
        if not 0.5 < betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"PointRowAdam: betas {betas} not supported")
        # always_rows: rows that may receive a gradient without being listed -- the confidence regulariser reads point 0
        # through every unfilled neighbour slot (studio_utils.py:193-199)
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0))
        me = weakref.ref(self)
        for group in self.param_groups:
            for p in group["params"]:
                _SUBSCRIBED[p] = me
        self._listed = WeakTensorKeyDictionary()    # parameters whose current gradient came with a row list
        self._always = tuple(int(r) for r in always_rows)
        self._ever: Dict[Tuple[int, str], _EverRows] = {}
        self._rebuild = False      # load_state_dict: recover the ever-touched set from the loaded second moments
        self._lib = None
        self.dense_steps = 0       # (diagnostics / tests) parameter updates that had to sweep every row

    # ---- state --------------------------------------------------------------------------------------------
    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)          # (torch.optim.Adam keeps it on the host too)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def _ever_for(self, p) -> _EverRows:
        n, _ = _rows_of(p)
        key = (n, str(p.device))
        ev = self._ever.get(key)
        if ev is None:
            ev = self._ever[key] = _EverRows(n, p.device)
            rows = [r for r in self._always if 0 <= r < n]
            if rows:
                ev.merge(self._lib, torch.tensor(rows, dtype=torch.int32, device=p.device), None,
                         C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
            # loaded moments (load_state_dict): every row with a non-zero second moment has had a gradient
            for group in self.param_groups if self._rebuild else ():
                for q in group["params"]:
                    st = self.state.get(q)
                    if st and "exp_avg_sq" in st and _rows_of(q)[0] == n and q.device == p.device:
                        width = _rows_of(q)[1]
                        hot = (st["exp_avg_sq"].reshape(n, width) != 0).any(dim=1).nonzero().reshape(-1).to(torch.int32)
                        if hot.numel():
                            ev.merge(self._lib, hot, None, C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
        return ev

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._ever = {}            # rebuilt from the loaded second moments at the next step (one O(N) pass)
        self._rebuild = True

    def _ingest(self, p, index: torch.Tensor, count: Optional[torch.Tensor], done: set) -> None:
        """publish_rows: the rows a backward wrote into p.grad -> the ever-touched set of p's row count, at once."""
        if not p.is_cuda:
            return
        if self._lib is None:
            self._lib = _lib.load()
        ev = self._ever_for(p)
        self._listed[p] = True
        if ev.dense or (id(ev), id(index)) in done:     # (the tensors of a backward share one list: merged once)
            return
        done.add((id(ev), id(index)))
        with torch.cuda.device(p.device):
            ev.merge(self._lib, index, count, C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))

    # ---- the step -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._lib is None:
            self._lib = _lib.load()
        lib = self._lib
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            # parameters of one row count and step count go out in one launch
            batches: Dict[Tuple[int, str, int], list] = {}
            for p in group["params"]:
                listed = self._listed.pop(p, False)
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("PointRowAdam: parameters must live on the GPU (the HIP path has no CPU fallback)")
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("PointRowAdam: float32 contiguous parameters with strided gradients only")
                st = self._state_of(p)
                st["step"] += 1
                ev = self._ever_for(p)
                if not listed and not ev.dense:
                    ev.dense = True          # a gradient nobody listed rows for: every row, from now on
                batches.setdefault((ev.num_rows, str(p.device), int(st["step"])), []).append((p, st, ev))
            for (num_rows, _, step), items in batches.items():
                bc1 = 1.0 - beta1 ** step
                bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
                step_size = lr / bc1
                ev = items[0][2]
                dev = items[0][0].device
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                if ev.dense:
                    self.dense_steps += len(items)
                for i in range(0, len(items), _lib.ADAM_MAX_TENSORS):
                    chunk = items[i:i + _lib.ADAM_MAX_TENSORS]
                    arr = (_lib.AdamTensorC * len(chunk))()
                    keep = []
                    for k, (p, st, _) in enumerate(chunk):
                        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                        keep.append(g)
                        arr[k].d_param, arr[k].d_grad = p.data_ptr(), g.data_ptr()
                        arr[k].d_exp_avg, arr[k].d_exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        arr[k].width = _rows_of(p)[1]
                    with torch.cuda.device(dev):
                        if ev.dense:
                            rc = lib.pnr_adam_rows(arr, len(chunk), num_rows, None, num_rows, None, beta1, beta2, eps,
                                                   step_size, bc2_sqrt, stream)
                        else:
                            rc = lib.pnr_adam_rows(arr, len(chunk), num_rows, ev.rows.data_ptr(), num_rows,
                                                   ev.count.data_ptr(), beta1, beta2, eps, step_size, bc2_sqrt, stream)
                    _lib.check(rc, "pnr_adam_rows")
        return loss

    def ever_touched(self) -> Dict[int, int]:
        """{row count: rows in the ever-touched set} (reads the device counts: diagnostics and tests only)."""
        return {ev.num_rows: (ev.num_rows if ev.dense else int(ev.count.item())) for ev in self._ever.values()}
