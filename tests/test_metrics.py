"""Image metrics behind PointNerf.get_image_metrics_and_images (reference studio_model.py:40-53,433-464): the key set
of the reference, closed forms against independent restatements (numpy / scipy.ndimage), and the LPIPS-free
state_dict / load_state_dict pair (studio_model.py:240-255)."""
import math

import numpy as np
import pytest
import torch
from scipy import ndimage

from pointnerf2studio_amd import metrics
from test_plugin_surface import _cpu_model


def _images(H=40, W=52, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(H, W, 3, generator=g)
    b = (a + 0.1 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    return a, b


def _skimage_ssim_restated(x, y, win=11, R=1.0):
    """skimage.metrics.structural_similarity (uniform window) per channel with scipy.ndimage, [H,W,C] float64."""
    vals = []
    npix = win * win
    cov = npix / (npix - 1.0)
    pad = (win - 1) // 2
    for c in range(x.shape[2]):
        X, Y = x[..., c], y[..., c]
        f = lambda t: ndimage.uniform_filter(t, size=win)
        ux, uy = f(X), f(Y)
        vx, vy, vxy = cov * (f(X * X) - ux * ux), cov * (f(Y * Y) - uy * uy), cov * (f(X * Y) - ux * uy)
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def test_closed_form_metrics_against_restatements():
    a, b = _images()
    A, B = torch.moveaxis(a, -1, 0)[None], torch.moveaxis(b, -1, 0)[None]
    mse = float(((a.double() - b.double()) ** 2).mean())
    assert abs(float(metrics.psnr(A, B)) - 10 * math.log10(1.0 / mse)) < 1e-4
    assert abs(float(metrics.rmse(A, B)) - math.sqrt(mse)) < 1e-9
    want = _skimage_ssim_restated(a.double().numpy(), b.double().numpy())
    assert abs(float(metrics.ssim_uniform(A, B)) - want) < 1e-9
    # identical images: every index is exactly 1; the gaussian form is symmetric and below 1 for different images
    assert float(metrics.ssim_uniform(A, A)) == pytest.approx(1.0, abs=1e-12)
    g = float(metrics.ssim_gaussian(A, B))
    assert float(metrics.ssim_gaussian(A, A)) == pytest.approx(1.0, abs=1e-6)
    assert 0.0 < g < 1.0 and g == pytest.approx(float(metrics.ssim_gaussian(B, A)), abs=1e-6)


def test_get_image_metrics_and_images_keys_and_shapes():
    m = _cpu_model()
    a, b = _images(H=24, W=30)
    outputs = {"coarse_raycolor": b.reshape(-1, 3)}
    md, im = m.get_image_metrics_and_images(outputs, {"image": a})
    assert set(md) == {"psnr", "skimage_ssim", "torchmetrics_ssim", "lpips", "lpips_vgg", "rmse"}   # studio_model.py:452-459
    assert all(isinstance(v, float) for v in md.values())
    assert set(im) == {"img"} and im["img"].shape == (24, 60, 3)            # ground truth | render, side by side
    assert torch.equal(im["img"][:, :30], a) and torch.equal(im["img"][:, 30:], b)
    assert outputs["ray_masked_coarse_raycolor"].shape == (24, 30, 3)
    assert md["psnr"] > 10 and 0 < md["rmse"] < 0.2 and 0 < md["skimage_ssim"] < 1
    if not metrics.HAVE_TORCHMETRICS:
        assert math.isnan(md["lpips"]) and math.isnan(md["lpips_vgg"])


def test_state_dict_never_holds_lpips_and_loads_strict():
    m = _cpu_model()
    m.lpips.register_buffer("probe", torch.ones(2))      # stands for the pretrained LPIPS tensors
    sd = m.state_dict()
    assert not [k for k in sd if k.startswith("lpips.") or k.startswith("lpips_vgg.")]
    assert "mlp_base.layers.0.weight" in sd and "neural_points.points_embeding" in sd
    res = m.load_state_dict(sd, strict=True)              # the missing lpips.* keys are filled from the module itself
    assert not res.missing_keys and not res.unexpected_keys
    # prefixed form (a pipeline saving `_model.`-prefixed keys, as nerfstudio does)
    assert not [k for k in m.state_dict(prefix="_model.") if "lpips" in k]
