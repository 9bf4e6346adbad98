"""A consumer must never read workspace rows beyond the counts the device holds.  Round 3 had such a reader: the plugin's
torch form of the confidence regulariser indexed `points_conf` with neighbour ids taken from rows of the render workspace
past `samples_selected` -- whatever an earlier call had left there -- and aborted the process with a device-side assert at
the next synchronisation (gpurun_out/r3_tests_1.log; fixed by masking before indexing).  Nothing
poisoned the workspaces then, so the class of bug had no test.  Here every workspace a training step and an eval render
use -- render workspace, training workspace (tape), the regulariser's scratch -- is filled with a byte pattern right
before the call: 0x7F (huge finite floats, neighbour ids far beyond N) and 0xFF (NaN floats, negative ids).  The step's
loss, image and every gradient must equal, bit for bit, those of a model whose workspaces were never touched."""
import pytest
import torch

import trajectory as T

pytestmark = pytest.mark.gpu


def _step(model, prob, view, device):
    model.zero_grad(set_to_none=True)
    out = model(T._bundle(prob["views"][view], device))
    losses = model.get_loss_dict(out, {"image": prob["views"][view]["target"].to(device)})
    sum(losses.values()).backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return {k: v.detach().clone() for k, v in losses.items()}, out["coarse_raycolor"].detach().clone(), grads


def _poison(model, byte):
    n = 0
    for rnd in model._renderers.values():
        for name in ("_ws", "_tws", "_conf_scratch"):
            t = getattr(rnd, name, None)
            if t is not None:
                t.fill_(byte)
                n += t.numel()
    return n


@pytest.mark.parametrize("conf_kernel", [True, False])
@pytest.mark.parametrize("byte", [0x7F, 0xFF])
def test_training_step_ignores_what_the_workspaces_held(oracle, gpu_device, byte, conf_kernel):
    prob = T.make_problem(oracle, N=30000, H=24, W=24)
    res = {}
    for poisoned in (False, True):
        model = T.make_model(prob, gpu_device)
        model.config.hip_conf_loss_kernel = conf_kernel
        model.train()
        model._render_calls = 0
        _step(model, prob, 0, gpu_device)              # allocates the workspaces; view 0 selects other samples than view 1
        if poisoned:
            assert _poison(model, byte) > 100e6         # render + training workspace of the worst-case capacity
            assert model._renderer_train.tape and model._renderer_train._tws is not None
        res[poisoned] = _step(model, prob, 1, gpu_device)
        torch.cuda.synchronize()
    (l0, rgb0, g0), (l1, rgb1, g1) = res[False], res[True]
    assert set(l0) == set(l1) == {"ray_masked_coarse_raycolor_loss", "conf_coefficient_loss"}
    for k in l0:
        assert torch.isfinite(l1[k]).all() and torch.equal(l0[k], l1[k]), f"{k}: {l0[k].item()} vs {l1[k].item()}"
    assert torch.equal(rgb0, rgb1)
    assert set(g0) == set(g1) and "neural_points.points_conf" in g0 and "neural_points.points_embeding" in g0
    for n in g0:
        assert torch.isfinite(g1[n]).all(), n
        assert torch.equal(g0[n], g1[n]), f"{n}: differs by {(g0[n] - g1[n]).abs().max().item():.3e} after poisoning"


@pytest.mark.parametrize("byte", [0x7F, 0xFF])
def test_eval_render_and_second_backward_ignore_poison(oracle, gpu_device, byte):
    """The eval path (its own renderer and workspace) and, on the training renderer, a render whose workspace was poisoned
    followed by TWO backwards (the tape must survive the first)."""
    prob = T.make_problem(oracle, N=30000, H=24, W=24)
    model = T.make_model(prob, gpu_device)
    clean = T.hip_eval_images(model, prob, gpu_device)
    _poison(model, byte)
    again = T.hip_eval_images(model, prob, gpu_device)
    for a, b in zip(clean, again):
        assert torch.equal(a, b)
    model.train()
    _step(model, prob, 0, gpu_device)
    _poison(model, byte)
    model.zero_grad(set_to_none=True)
    out = model(T._bundle(prob["views"][1], gpu_device))
    loss = (out["coarse_raycolor"] * prob["views"][1]["target"].to(gpu_device)).sum()
    loss.backward(retain_graph=True)
    first = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all() and torch.equal(p.grad, first[n]), n
