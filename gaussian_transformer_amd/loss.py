"""Image losses that produce dL/dimage for the rasterizer backward.

Restates utils/loss_utils.py:17-63 (l1_loss, ssim: 11x11 Gaussian window sigma 1.5,
C1=0.01^2, C2=0.03^2, zero padding 5, grouped conv) and utils/image_utils.py:17-19 (psnr),
and the training objective of train.py:91-92 with lambda_dssim = 0.2
(arguments/__init__.py:83).  Plain torch ops (MIOpen/rocBLAS on ROCm); SURVEY.md 8a-1 marks
them "reuse torch, do not rewrite".  Pinned by tests/golden/loss.npz.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LAMBDA_DSSIM = 0.2

_window_cache = {}


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return ((network_output - gt) ** 2).mean()


def _window(window_size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    key = (window_size, channel, like.device, like.dtype)
    w = _window_cache.get(key)
    if w is None:
        g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
        w = w2.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)
        _window_cache[key] = w
    return w


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    channel = img1.size(-3)
    w = _window(window_size, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, w, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, w, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, w, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, w, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def training_loss(image: torch.Tensor, gt_image: torch.Tensor, lambda_dssim: float = LAMBDA_DSSIM) -> torch.Tensor:
    """(1-lambda) L1 + lambda (1 - SSIM): train.py:91-92."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt_image) + lambda_dssim * (1.0 - ssim(image, gt_image))
