"""optim.PointRowAdam (pnr_rows_merge + pnr_adam_rows) against torch.optim.Adam, the optimiser the reference registers for
the `neural_points` group (studio_config.py:41-47: AdamOptimizerConfig(lr=0.002) -> torch.optim.Adam, eps 1e-8).  Adam over
the rows that ever had a gradient IS dense Adam (a never-touched row has zero moments and moves by exactly 0): the tests
hold parameters to 1e-7 absolute and the moments to fp32 rounding over 50 steps, rows that were never touched to their
initial bits, and the state dict to be interchangeable with torch's."""
import copy

import pytest
import torch

import trajectory as T
from pointnerf2studio_amd.optim import PointRowAdam, publish_rows

pytestmark = pytest.mark.gpu

SHAPES = [(1, 32), (1, 3), (1, 3), (1, 1)]      # embedding, color, dir, conf: [1, N, C]


def _params(N, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn((1, N, c), generator=g).to(device)) for _, c in SHAPES]


def _sparse_grads(params, N, n_rows, gen, device, pad=257):
    """Dense gradient tensors that are zero outside `n_rows` random rows + the padded device row list and count a fused
    backward would publish (entries behind the count repeat the first one, as pnr_render_touched pads)."""
    rows = torch.randperm(N, generator=gen)[:n_rows].sort()[0]
    grads = []
    for p in params:
        gfull = torch.zeros_like(p)
        gfull[0, rows.to(device)] = (torch.randn((n_rows, p.shape[-1]), generator=gen) *
                                     10.0 ** float(torch.randint(-6, 1, (1,), generator=gen))).to(device)
        grads.append(gfull)
    index = torch.cat([rows, rows[:1].expand(pad)]).to(device=device, dtype=torch.int32)
    count = torch.tensor([n_rows], dtype=torch.int64, device=device)
    return grads, index, count, rows


def test_fifty_steps_equal_torch_adam(gpu_device):
    N, dev = 200_000, gpu_device
    ours, theirs = _params(N, dev), _params(N, dev)
    init = [p.detach().clone() for p in ours]
    opt_a = PointRowAdam(ours, lr=2e-3, eps=1e-8)
    opt_b = torch.optim.Adam(theirs, lr=2e-3, eps=1e-8)
    sch_a = torch.optim.lr_scheduler.LambdaLR(opt_a, T.lr_lambda)      # the reference's schedule (studio_utils.py:33-44)
    sch_b = torch.optim.lr_scheduler.LambdaLR(opt_b, T.lr_lambda)
    gen = torch.Generator().manual_seed(1)
    union = set()
    for step in range(50):
        grads, index, count, rows = _sparse_grads(ours, N, 1500 + 37 * step, gen, dev)
        union.update(rows.tolist())
        for p, q, g in zip(ours, theirs, grads):
            p.grad, q.grad = g, g.clone()
        publish_rows(ours, index, count)
        opt_a.step()
        opt_b.step()
        sch_a.step()
        sch_b.step()
    assert opt_a.dense_steps == 0 and opt_a.ever_touched() == {N: len(union | {0})}
    never = torch.ones(N, dtype=torch.bool)
    never[torch.tensor(sorted(union | {0}))] = False
    for p, q, p0 in zip(ours, theirs, init):
        assert (p - q).abs().max().item() <= 1e-7, f"{tuple(p.shape)}: {(p - q).abs().max().item():.3e}"
        assert torch.equal(p[0, never.to(dev)], p0[0, never.to(dev)]) and torch.equal(q[0, never.to(dev)], p0[0, never.to(dev)])
        sa, sb = opt_a.state[p], opt_b.state[q]
        assert float(sa["step"]) == float(sb["step"]) == 50.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(sa[k], sb[k], rtol=2e-6, atol=0.0), k
            assert (sa[k] != 0).reshape(N, -1).any(1).sum().item() <= len(union)


def test_unlisted_gradient_falls_back_to_a_dense_sweep(gpu_device):
    """A gradient nobody published rows for (another backward path) is applied densely: exact, merely not sparse."""
    N, dev = 50_000, gpu_device
    ours, theirs = _params(N, dev, 3), _params(N, dev, 3)
    opt_a, opt_b = PointRowAdam(ours, lr=1e-3), torch.optim.Adam(theirs, lr=1e-3)
    gen = torch.Generator().manual_seed(4)
    for step in range(5):
        for p, q in zip(ours, theirs):
            g = torch.randn(p.shape, generator=gen).to(dev)
            p.grad, q.grad = g, g.clone()
        opt_a.step()
        opt_b.step()
    assert opt_a.dense_steps == 5 * len(ours) and opt_a.ever_touched() == {N: N}
    for p, q in zip(ours, theirs):
        assert (p - q).abs().max().item() <= 1e-7


def test_state_dict_is_interchangeable_with_torch_adam(gpu_device):
    N, dev = 60_000, gpu_device
    ours, theirs = _params(N, dev, 5), _params(N, dev, 5)
    opt_a, opt_b = PointRowAdam(ours, lr=2e-3), torch.optim.Adam(theirs, lr=2e-3)
    gen = torch.Generator().manual_seed(6)

    def steps(a, b, pa, pb, n):
        for _ in range(n):
            grads, index, count, _ = _sparse_grads(pa, N, 800, gen, dev)
            for p, q, g in zip(pa, pb, grads):
                p.grad, q.grad = g, g.clone()
            publish_rows(pa, index, count)
            a.step()
            b.step()
    steps(opt_a, opt_b, ours, theirs, 10)
    # cross-load: torch's state into a new PointRowAdam and ours into a new torch Adam, then 10 more steps each
    ours2 = [torch.nn.Parameter(p.detach().clone()) for p in theirs]
    theirs2 = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    opt_a2, opt_b2 = PointRowAdam(ours2, lr=2e-3), torch.optim.Adam(theirs2, lr=2e-3)
    opt_a2.load_state_dict(copy.deepcopy(opt_b.state_dict()))
    opt_b2.load_state_dict(copy.deepcopy(opt_a.state_dict()))
    steps(opt_a2, opt_b2, ours2, theirs2, 10)
    assert opt_a2.dense_steps == 0
    ever = opt_a2.ever_touched()[N]
    assert 800 < ever <= 20 * 800 + 1          # recovered from the loaded second moments, not "every row"
    for p, q in zip(ours2, theirs2):
        assert (p - q).abs().max().item() <= 2e-7


def test_plugin_training_with_point_row_adam_equals_torch_adam(oracle, gpu_device):
    """The training loop of tests/trajectory.py with the optimisers as studio_config registers them ("fields": torch Adam,
    "neural_points": PointRowAdam) against the same loop with torch.optim.Adam for both groups: 25 steps, the same losses
    and parameters; the sparse optimiser never sweeps."""
    prob = T.make_problem(oracle, N=30000, H=24, W=24)
    runs = {}
    for sparse in (False, True):
        model = T.make_model(prob, gpu_device)
        model.train()
        groups = model.get_param_groups()
        opt_f = torch.optim.Adam(groups["fields"], lr=T.LR["fields"], eps=1e-8)
        opt_p = (PointRowAdam if sparse else torch.optim.Adam)(groups["neural_points"], lr=T.LR["neural_points"], eps=1e-8)
        callbacks = model.get_training_callbacks(None)
        targets = [v["target"].to(gpu_device) for v in prob["views"]]
        losses = []
        for i in range(25):
            k = i % 2
            opt_f.zero_grad(set_to_none=True)
            opt_p.zero_grad(set_to_none=True)
            out = model(T._bundle(prob["views"][k], gpu_device))
            loss = sum(model.get_loss_dict(out, {"image": targets[k]}).values())
            loss.backward()
            opt_f.step()
            opt_p.step()
            for cb in callbacks:
                cb.run_callback(step=i)
            losses.append(loss.detach())
        runs[sparse] = ([float(x) for x in torch.stack(losses).cpu()], T.hip_state(model), opt_p)
    la, lb = runs[True][0], runs[False][0]
    assert max(abs(a - b) / abs(b) for a, b in zip(la, lb)) <= 1e-5, (la, lb)
    opt = runs[True][2]
    assert opt.dense_steps == 0
    n_ever = opt.ever_touched()[prob["points"]["xyz"].shape[0]]
    assert 1000 < n_ever < 0.6 * prob["points"]["xyz"].shape[0]
    for part_a, part_b in zip(runs[True][1], runs[False][1]):
        for k in part_a:
            d = (part_a[k] - part_b[k]).abs().max().item()
            assert d <= 2e-5 * max(part_b[k].abs().max().item(), 1.0), f"{k}: {d:.3e}"
