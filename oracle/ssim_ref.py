"""CPU restatement (numpy, float64 arithmetic on the float32 window) of the training loss
(1-lambda) L1 + lambda (1-SSIM) and its gradient w.r.t. the rendered image.

TEST INFRASTRUCTURE ONLY.  Follows utils/loss_utils.py:17-63 of the reference (11x11 window = outer
product of the normalised float32 1-D Gaussian sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean
over all C*H*W entries) and train.py:91-92.  PINNED by tests/golden/loss.npz, which holds the reference's
own l1/ssim/loss values and autograd gradient (oracle/make_golden.py).  Uses the full 2-D window, not the
separable form of the HIP kernel.
"""
import math

import numpy as np

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window(size=11, sigma=1.5):
    g = np.array([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], dtype=np.float32)
    g = g / g.sum(dtype=np.float32)
    return (g[:, None] * g[None, :]).astype(np.float32).astype(np.float64)


def _corr(a, w):
    """zero-padded 'same' correlation of [C,H,W] with the 2-D window"""
    r = w.shape[0] // 2
    p = np.pad(a, ((0, 0), (r, r), (r, r)))
    out = np.zeros_like(a, dtype=np.float64)
    H, W = a.shape[1:]
    for i in range(w.shape[0]):
        for j in range(w.shape[1]):
            out += w[i, j] * p[:, i:i + H, j:j + W]
    return out


def l1_ssim_loss(img, gt, lam=0.2, with_grad=True):
    x = np.asarray(img, dtype=np.float64); y = np.asarray(gt, dtype=np.float64)
    w = window()
    m1, m2 = _corr(x, w), _corr(y, w)
    s11, s22, s12 = _corr(x * x, w), _corr(y * y, w), _corr(x * y, w)
    a1 = 2 * m1 * m2 + C1; a2 = 2 * (s12 - m1 * m2) + C2
    b1 = m1 * m1 + m2 * m2 + C1; b2 = (s11 - m1 * m1) + (s22 - m2 * m2) + C2
    f = a1 * a2 / (b1 * b2)
    n = x.size
    l1 = np.abs(x - y).mean(); ssim = f.mean()
    loss = (1 - lam) * l1 + lam * (1 - ssim)
    if not with_grad:
        return loss, l1, ssim, None
    D = b1 * b2
    A = (2 * m2 * (a2 - a1) - f * 2 * m1 * (b2 - b1)) / D       # d f / d mu1, with sigma terms expressed through mu1
    B = -f / b2                                                 # d f / d E[x^2]
    Cm = 2 * a1 / D                                             # d f / d E[xy]
    dssim = _corr(A, w) + 2 * x * _corr(B, w) + y * _corr(Cm, w)  # window symmetric: adjoint of corr == corr
    grad = ((1 - lam) * np.sign(x - y) - lam * dssim) / n
    return loss, l1, ssim, grad
