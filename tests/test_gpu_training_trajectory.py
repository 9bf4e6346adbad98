"""N optimiser steps of the training path next to the same N steps of autograd through the CPU oracle ("matched PSNR"):
every other training test differentiates ONE step.  What only shows over many steps -- packed point rows refreshed from
the bound parameters by each render, the packed weights updated after each Adam step, the dense `.grad` buffers handed
out again and cleaned by the row lists of EARLIER steps, the tape workspace reused -- is what the reference drives 200 000
times (studio_model.py:415-431, studio_config.py:17,33-48).

A teacher network renders two 32 x 32 target images of a 40 k-point scene through the oracle; a student (other weights,
other colours, perturbed embeddings) is fitted to them for 50 Adam steps at the reference's learning rates and 0.3 jitter,
(a) through PointNerf.forward + get_loss_dict + backward + the callbacks on the HIP path, (b) through torch autograd over
oracle.render on the CPU with the same jitter uniforms.  What "the same" can mean over 50 Adam steps is measured, not
assumed: the oracle is also run from embeddings half an ulp away, and the HIP path must stay as close to the oracle as the
oracle stays to that copy of itself (x4), with floors of 1e-5 / 1e-3 on the loss where no amplification is needed to explain
a difference."""
import pytest
import torch

import trajectory as T

pytestmark = pytest.mark.gpu

STEPS = 50


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def _rel_l2(a, b):
    return ((a - b).double().norm() / b.double().norm().clamp(min=1e-30)).item()


def _self_sensitivity(oracle, prob, steps, seeds, losses_o, pts_o, w_o, edit_at=None):
    """How far the ORACLE drifts from itself when the student's embeddings start half an ulp away (1e-7 relative): Adam steps
    through LeakyReLU kinks and an alpha composite amplify rounding-level differences, most of all right after a prune + grow
    (measured: 4e-3 of the loss over 50 plain steps, 2.5e-2 in the steps behind the edit).  The HIP path -- another float32
    evaluation order of the same function -- cannot be asked to stay closer to the oracle than the oracle stays to itself."""
    l2, p2, w2 = T.run_oracle(oracle, T.perturbed(prob), steps, seeds, edit_at=edit_at)
    drift = [_rel(a, b) for a, b in zip(l2, losses_o)]
    par = max([_rel_l2(p2[n], pts_o[n]) for n in T.POINT_KEYS] + [_rel_l2(w2[n], t) for n, t in w_o.items()])
    return drift, par


def _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, device, sens, floors, transient=None, psnr_tol=0.1):
    """sens: (per-step loss drift, parameter L2) of the oracle against its perturbed self; floors: (first steps, all steps,
    parameter L2) below which no sensitivity is needed to explain a difference."""
    drift = [_rel(h, o) for h, o in zip(losses_h, losses_o)]
    print("per-step loss drift, HIP vs oracle:   ", " ".join(f"{d:.1e}" for d in drift))
    print("per-step loss drift, oracle vs itself:", " ".join(f"{d:.1e}" for d in sens[0]))
    first_tol = max(floors[0], 4.0 * max(sens[0][:10]))
    assert max(drift[:10]) <= first_tol, f"first 10 steps: {max(drift[:10]):.3e} (oracle vs itself: {max(sens[0][:10]):.3e})"
    # `transient` = (first, last) steps behind a prune + grow.  The loss there is carried by the few rays the new, untrained
    # points sit on, and rounding-level differences grow step by step instead of staying put (oracle vs its perturbed self:
    # 2.5e-2 on one host, 4e-3 on another; HIP vs oracle 3e-3 with one build of the backward, 1e-2 .. 5e-2 with the next, whose
    # gradients differ in the seventh digit).  What a state-handling bug would do -- a jump AT the edit, per cent of the loss or
    # more from the first step on the new cloud -- is excluded by the steps up to and including that first one, held to the
    # oracle's own drift; behind it only the size of the deviation is bounded (mean 5 %, any step 25 %)
    lo, hi = transient if transient else (len(drift), len(drift))
    calm = drift[:lo] + drift[hi:]
    calm_sens = sens[0][:lo] + sens[0][hi:]
    last_tol = max(floors[1], 4.0 * max(calm_sens))
    assert max(calm) <= last_tol, f"{len(calm)} steps outside the transient: {max(calm):.3e} (oracle vs itself: {max(calm_sens):.3e})"
    if hi > lo:
        assert max(drift[lo:hi]) <= 0.25 and sum(drift[lo:hi]) / (hi - lo) <= 0.05, \
            f"behind the edit: max {max(drift[lo:hi]):.3e}, mean {sum(drift[lo:hi]) / (hi - lo):.3e}"
    pts_h, w_h = T.hip_state(model)
    # parameters: Adam normalises every element's step to ~lr whatever the size of its gradient, so an element whose
    # gradient is rounding noise around zero walks +-lr per step on either side: single entries may sit up to steps x lr
    # apart while the tensors agree in the mean.  The relative L2 distance of every tensor is held (against the oracle's own
    # sensitivity), the largest single deviation only to the Adam bound
    worst_l2, worst_max, who = 0.0, 0.0, ("", "")
    pairs = [(n, pts_h[n], pts_o[n]) for n in T.POINT_KEYS] + [(n, w_h[n], t) for n, t in w_o.items()]
    for name, a, b in pairs:
        l2 = _rel_l2(a, b)
        mx = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        if l2 > worst_l2:
            worst_l2, who = l2, (name, who[1])
        if mx > worst_max:
            worst_max, who = mx, (who[0], name)
    l2_tol = max(floors[2], 4.0 * sens[1])
    assert worst_l2 <= l2_tol, f"parameters after {len(losses_h)} steps: relative L2 {worst_l2:.3e} ({who[0]}; oracle vs itself {sens[1]:.3e})"
    assert worst_max <= 2.0 * len(losses_h) * T.LR["neural_points"] / 0.5, f"largest deviation {worst_max:.3e} ({who[1]})"
    worst = (worst_l2, worst_max, who)
    # matched PSNR: eval images (jitter 0, clamp) of the two trained students against the teacher's
    img_h = T.hip_eval_images(model, prob, device)
    img_o = T.eval_images(oracle, prob, pts_o, w_o)
    ps_h = [T.psnr(a, v["target"]) for a, v in zip(img_h, prob["views"])]
    ps_o = [T.psnr(a, v["target"]) for a, v in zip(img_o, prob["views"])]
    for a, b in zip(ps_h, ps_o):
        assert abs(a - b) <= psnr_tol, f"PSNR vs teacher: HIP {ps_h} dB, oracle {ps_o} dB"
    return drift, worst, ps_h, ps_o


def test_fifty_training_steps_follow_the_oracle(oracle, gpu_device):
    prob = T.make_problem(oracle)
    ps_0 = [T.psnr(a, v["target"]) for a, v in zip(T.eval_images(oracle, prob, prob["points"], prob["weights"]),
                                                    prob["views"])]
    losses_h, seeds, model = T.run_hip(prob, STEPS, gpu_device)
    assert seeds == list(range(STEPS)) and model.host_reads <= 1    # (the collider's planes, once)
    losses_o, pts_o, w_o = T.run_oracle(oracle, prob, STEPS, seeds)
    sens = _self_sensitivity(oracle, prob, STEPS, seeds, losses_o, pts_o, w_o)
    drift, worst, ps_h, ps_o = _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, gpu_device, sens,
                                        floors=(1e-5, 1e-3, 1e-3))
    print(f"oracle vs its perturbed self: loss {max(sens[0][:10]):.2e} / {max(sens[0]):.2e}, parameters L2 {sens[1]:.2e}")
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
    sens = _self_sensitivity(oracle, prob, 40, seeds, losses_o, pts_o, w_o, edit_at=20)
    drift, worst, ps_h, ps_o = _compare(oracle, prob, losses_h, model, losses_o, pts_o, w_o, gpu_device, sens,
                                        floors=(1e-5, 1e-3, 5e-3), transient=(21, 40), psnr_tol=0.3)
    # (steps 0..19 on the old cloud and step 20, the first one on the edited cloud, are the `calm` ones above)
    print(f"oracle vs its perturbed self: loss {max(sens[0][:20]):.2e} before / {max(sens[0]):.2e} behind the edit, "
          f"parameters L2 {sens[1]:.2e}")
    print(f"prune + grow: loss drift {max(drift):.2e}, parameters L2 {worst[0]:.2e} ({worst[2][0]}) / largest entry {worst[1]:.2e} ({worst[2][1]}), PSNR HIP {ps_h} / oracle {ps_o} dB")
