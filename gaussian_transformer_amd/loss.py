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


class _FusedL1SSIM(torch.autograd.Function):
    """(1-lambda) L1 + lambda (1-SSIM) in one HIP kernel per direction (include/gsr_loss.h)."""

    @staticmethod
    def forward(ctx, image, gt_image, lambda_dssim):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        dev = image.device
        if dev.type != "cuda":
            raise _lib.GsrError(f"fused_l1_ssim_loss needs tensors on a HIP device, got {dev} (use training_loss for the torch path)")
        if image.dtype != torch.float32 or gt_image.dtype != torch.float32 or image.shape != gt_image.shape or image.dim() != 3:
            raise _lib.GsrError("fused_l1_ssim_loss: image and gt_image must be float32 [C,H,W] tensors of equal shape")
        img = image.contiguous(); gt = gt_image.to(dev).contiguous()
        Cn, H, W = (int(x) for x in img.shape)
        nb = C.c_size_t()
        _lib.check(lib.gsr_l1_ssim_workspace(Cn, H, W, C.byref(nb)), "gsr_l1_ssim_workspace")
        with torch.cuda.device(dev):
            ws = torch.empty((nb.value,), dtype=torch.uint8, device=dev)
            out = torch.empty((3,), dtype=torch.float32, device=dev)
            _lib.check(lib.gsr_l1_ssim_forward(torch.cuda.current_stream(dev).cuda_stream, Cn, H, W, img.data_ptr(), gt.data_ptr(),
                                               float(lambda_dssim), out.data_ptr(), ws.data_ptr(), nb.value), "gsr_l1_ssim_forward")
        ctx.save_for_backward(img, gt, ws)
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.terms = out
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        from . import _lib
        lib = _lib.load()
        img, gt, ws = ctx.saved_tensors
        dev = img.device
        Cn, H, W = (int(x) for x in img.shape)
        with torch.cuda.device(dev):
            g = grad_loss.to(device=dev, dtype=torch.float32).contiguous().reshape(1)
            grad = torch.empty_like(img)
            _lib.check(lib.gsr_l1_ssim_backward(torch.cuda.current_stream(dev).cuda_stream, Cn, H, W, img.data_ptr(), gt.data_ptr(),
                                                ctx.lambda_dssim, g.data_ptr(), ws.data_ptr(), ws.numel(), grad.data_ptr()),
                       "gsr_l1_ssim_backward")
        return grad, None, None


def fused_l1_ssim_loss(image: torch.Tensor, gt_image: torch.Tensor, lambda_dssim: float = LAMBDA_DSSIM) -> torch.Tensor:
    """Drop-in for training_loss() on a HIP device: same value and gradient w.r.t. `image`
    (gt_image gets no gradient, as in the reference's use at train.py:90-93)."""
    return _FusedL1SSIM.apply(image, gt_image, lambda_dssim)
