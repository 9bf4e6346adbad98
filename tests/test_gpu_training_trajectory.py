"""N optimiser steps of the training path next to the same N steps of autograd through the CPU oracle ("matched PSNR"):
every other training test differentiates ONE step.  What only shows over many steps -- packed point rows refreshed from
the bound parameters by each render, the packed weights updated after each Adam step, the dense `.grad` buffers handed
out again and cleaned by the row lists of EARLIER steps, the tape workspace reused -- is what the reference drives 200 000
times (studio_model.py:415-431, studio_config.py:17,33-48).

A teacher network renders two 32 x 32 target images of a 40 k-point scene through the oracle; a student (other weights,
other colours, perturbed embeddings) is fitted to them for 50 Adam steps at the reference's learning rates and 0.3 jitter,
(a) through PointNerf.forward + get_loss_dict + backward + the callbacks on the HIP path, (b) through torch autograd over
oracle.render on the CPU with the same jitter uniforms.  Tolerances state the measured drift."""
import pytest
import torch

import trajectory as T

pytestmark = pytest.mark.gpu

STEPS = 50


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, device, first_tol, last_tol, param_tol):
    # per-step loss: early steps see (almost) the same parameters; later ones carry what 50 Adam steps made of fp32
    # summation-order differences and of LeakyReLU units that sit within rounding of their kink
    drift = [_rel(h, o) for h, o in zip(losses_h, losses_o)]
    assert max(drift[:10]) <= first_tol, f"first 10 steps: {max(drift[:10]):.3e}"
    assert max(drift) <= last_tol, f"all {len(drift)} steps: {max(drift):.3e}"
    pts_h, w_h = T.hip_state(model)
    # parameters: Adam normalises every element's step to ~lr whatever the size of its gradient, so an element whose
    # gradient is rounding noise around zero (its sign decided by the summation order) walks +-lr per step on either side:
    # single entries may sit up to steps x lr apart while the tensors agree in the mean.  Both are held: the relative L2
    # distance of every tensor and its largest single deviation in units of the tensor's largest entry
    worst_l2, worst_max, who = 0.0, 0.0, ("", "")
    pairs = [(n, pts_h[n], pts_o[n]) for n in T.POINT_KEYS] + [(n, w_h[n], t) for n, t in w_o.items()]
    for name, a, b in pairs:
        l2 = ((a - b).double().norm() / b.double().norm().clamp(min=1e-30)).item()
        mx = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        if l2 > worst_l2:
            worst_l2, who = l2, (name, who[1])
        if mx > worst_max:
            worst_max, who = mx, (who[0], name)
    assert worst_l2 <= param_tol[0], f"parameters after {len(losses_h)} steps: relative L2 {worst_l2:.3e} ({who[0]})"
    assert worst_max <= param_tol[1], f"parameters after {len(losses_h)} steps: largest deviation {worst_max:.3e} ({who[1]})"
    worst = (worst_l2, worst_max, who)
    # matched PSNR: eval images (jitter 0, clamp) of the two trained students against the teacher's
    img_h = T.hip_eval_images(model, prob, device)
    img_o = T.eval_images(oracle, prob, pts_o, w_o)
    ps_h = [T.psnr(a, v["target"]) for a, v in zip(img_h, prob["views"])]
    ps_o = [T.psnr(a, v["target"]) for a, v in zip(img_o, prob["views"])]
    for a, b in zip(ps_h, ps_o):
        assert abs(a - b) <= 0.1, f"PSNR vs teacher: HIP {ps_h} dB, oracle {ps_o} dB"
    return drift, worst, ps_h, ps_o


def test_fifty_training_steps_follow_the_oracle(oracle, gpu_device):
    prob = T.make_problem(oracle)
    ps_0 = [T.psnr(a, v["target"]) for a, v in zip(T.eval_images(oracle, prob, prob["points"], prob["weights"]),
                                                    prob["views"])]
    losses_h, seeds, model = T.run_hip(prob, STEPS, gpu_device)
    assert seeds == list(range(STEPS)) and model.host_reads <= 1    # (the collider's planes, once)
    losses_o, pts_o, w_o = T.run_oracle(oracle, prob, STEPS, seeds)
    drift, worst, ps_h, ps_o = _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, gpu_device,
                                        first_tol=1e-4, last_tol=1e-2, param_tol=(1e-2, 1e-1))
    print(f"loss drift: first 10 steps {max(drift[:10]):.2e}, all {max(drift):.2e}; parameters: L2 {worst[0]:.2e} ({worst[2][0]}), largest entry {worst[1]:.2e} ({worst[2][1]}); "
          f"PSNR vs teacher {ps_0} -> HIP {ps_h} / oracle {ps_o} dB")
    assert losses_h[-1] < 0.1 * losses_h[0]
    for a, b in zip(ps_h, ps_0):
        assert a >= b + 3.0, f"training must gain at least 3 dB: {ps_0} -> {ps_h}"
    # ... and the run is repeatable bit for bit: a second model, same seeds, same kernels
    losses_2, seeds_2, model_2 = T.run_hip(prob, STEPS, gpu_device)
    assert seeds_2 == seeds and losses_2 == losses_h
    a, b = T.hip_state(model), T.hip_state(model_2)
    for part_a, part_b in zip(a, b):
        for k in part_a:
            assert torch.equal(part_a[k], part_b[k]), f"{k}: a repeated training run differs"


def test_training_through_a_prune_and_a_grow(oracle, gpu_device):
    """The same run with the cloud edited between steps 20 and 21: prune(conf < 0.25) then grow_points (1500 seeded
    points), optimisers re-created as the reference's trainer does (run/train_studio.py:676-684,714-716).  The voxel
    structure is updated in place (pnr_scene_update), the bound tensors, the dense gradient buffers and their pending row
    lists belong to a cloud that no longer exists."""
    prob = T.make_problem(oracle)
    losses_h, seeds, model = T.run_hip(prob, 40, gpu_device, edit_at=20)
    losses_o, pts_o, w_o = T.run_oracle(oracle, prob, 40, seeds, edit_at=20)
    assert model.neural_points.points_xyz.shape[0] == pts_o["xyz"].shape[0] != prob["points"]["xyz"].shape[0], \
        "the two sides pruned a different set (a confidence within rounding of the threshold)"
    assert torch.equal(model.neural_points.points_xyz.detach().cpu(), pts_o["xyz"])
    drift, worst, ps_h, ps_o = _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, gpu_device,
                                        first_tol=1e-4, last_tol=1e-2, param_tol=(1e-2, 1e-1))
    print(f"prune + grow: loss drift {max(drift):.2e}, parameters L2 {worst[0]:.2e} ({worst[2][0]}) / largest entry {worst[1]:.2e} ({worst[2][1]}), PSNR HIP {ps_h} / oracle {ps_o} dB")
