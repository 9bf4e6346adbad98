"""Image metrics of `PointNerf.get_image_metrics_and_images` (reference studio_model.py:40-53,226-237,433-464).

The reference computes them with third-party packages: torchmetrics (`PeakSignalNoiseRatio`,
`structural_similarity_index_measure`, `LearnedPerceptualImagePatchSimilarity` alex + vgg) and scikit-image
(`structural_similarity` with an 11-pixel uniform window, `mean_squared_error`).  Neither is installable in the build
image, so the closed-form ones are restated here with torch ops (they run on whatever device the images are on) and
the real packages are used whenever they import:

  psnr                10 log10(data_range^2 / mse)                          torchmetrics PeakSignalNoiseRatio(data_range=1)
  rmse                sqrt(mean((a - b)^2)) per image, mean over images      studio_model.py:48-53
  skimage_ssim        Wang et al. SSIM, 11 x 11 UNIFORM window, sample covariance (N / (N - 1)), data_range 1, mean
                      over the interior (window fully inside) and over channels   studio_model.py:40-46  [skimage-mem]
  torchmetrics_ssim   the same index with an 11 x 11 GAUSSIAN window (sigma 1.5), population covariance, data range
                      taken from the images (max - min over both), interior mean   [tm-mem]
  lpips, lpips_vgg    learned metrics: need the pretrained AlexNet / VGG weights torchmetrics downloads; without
                      torchmetrics they are reported as NaN (the keys stay, the surface stays)

[skimage-mem] / [tm-mem]: restated from the packages' published algorithms, un-pinned dependencies of the reference
(pyproject.toml lists no versions for them); tests/test_metrics.py checks the uniform-window form against a
scipy.ndimage restatement.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

try:  # pragma: no cover - not installable in the build image
    from torchmetrics.functional.image import structural_similarity_index_measure as _tm_ssim
    from torchmetrics.image.lpip import LearnedPerceptualImagePatchSimilarity as _TMLpips
    HAVE_TORCHMETRICS = True
except Exception:
    _tm_ssim, _TMLpips, HAVE_TORCHMETRICS = None, None, False


def psnr(image: torch.Tensor, rgb: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """[1,C,H,W] images -> scalar tensor."""
    mse = torch.mean((image.float() - rgb.float()) ** 2)
    return 10.0 * torch.log10(torch.as_tensor(data_range ** 2, device=mse.device) / mse)


def rmse(image: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
    """studio_model.py:48-53: sqrt of skimage's mean_squared_error per image (float64 accumulation), mean over images."""
    d = (image.double() - rgb.double()) ** 2
    return torch.sqrt(d.flatten(1).mean(dim=1)).mean()


def _window_ssim(x: torch.Tensor, y: torch.Tensor, kernel: torch.Tensor, data_range: float, cov_norm: float,
                 k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """SSIM map over the interior (no padding: only windows that lie fully inside the image), per channel."""
    C = x.shape[1]
    w = kernel.to(x)[None, None].expand(C, 1, -1, -1)
    filt = lambda t: F.conv2d(t, w, groups=C)
    ux, uy = filt(x), filt(y)
    uxx, uyy, uxy = filt(x * x), filt(y * y), filt(x * y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    return ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))


def ssim_uniform(image: torch.Tensor, rgb: torch.Tensor, win_size: int = 11, data_range: float = 1.0) -> torch.Tensor:
    """skimage.metrics.structural_similarity(win_size=11, channel_axis, data_range=1.0) per image, mean over images
    (studio_model.py:40-46).  [N,C,H,W] in, scalar out."""
    if min(image.shape[-2:]) < win_size:
        raise ValueError(f"win_size {win_size} exceeds the image extent {tuple(image.shape[-2:])}")
    n = win_size * win_size
    k = torch.full((win_size, win_size), 1.0 / n, dtype=torch.float64)
    s = _window_ssim(image.double(), rgb.double(), k, data_range, n / (n - 1.0))
    return s.flatten(2).mean(dim=2).mean(dim=1).mean()


def ssim_gaussian(image: torch.Tensor, rgb: torch.Tensor, kernel_size: int = 11, sigma: float = 1.5,
                  data_range: Optional[float] = None) -> torch.Tensor:
    """torchmetrics.functional.image.structural_similarity_index_measure with its defaults.  [N,C,H,W] -> scalar."""
    if HAVE_TORCHMETRICS:  # pragma: no cover
        return _tm_ssim(image, rgb)
    if data_range is None:
        data_range = float(torch.max(image.max() - image.min(), rgb.max() - rgb.min()))
    ax = torch.arange(kernel_size, dtype=torch.float64) - (kernel_size - 1) / 2.0
    g = torch.exp(-(ax / sigma) ** 2 / 2)
    g = g / g.sum()
    s = _window_ssim(image.double(), rgb.double(), torch.outer(g, g), data_range, 1.0)
    return s.flatten(1).mean(dim=1).mean().float()


class Lpips(torch.nn.Module):
    """LearnedPerceptualImagePatchSimilarity(net_type) when torchmetrics (and its pretrained weights) exist, else a
    parameter-free stand-in that reports NaN: the metric key stays, nothing is faked."""

    def __init__(self, net_type: str = "alex") -> None:
        super().__init__()
        self.net_type = net_type
        self.impl = None
        if HAVE_TORCHMETRICS:  # pragma: no cover
            try:
                self.impl = _TMLpips(net_type=net_type)
            except Exception:
                self.impl = None

    def forward(self, image: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
        if self.impl is None:
            return torch.tensor(math.nan)
        return self.impl(image, rgb)  # pragma: no cover
