"""-m gpu: fused L1 + SSIM loss kernel (include/gsr_loss.h) vs the reference's golden values
(tests/golden/loss.npz), the numpy oracle (oracle/ssim_ref.py) and the torch restatement."""
import os

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import loss
from oracle import ssim_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _run(img, gt, lam=0.2, up=1.0):
    a = torch.tensor(img, dtype=torch.float32, device="cuda", requires_grad=True)
    b = torch.tensor(gt, dtype=torch.float32, device="cuda")
    L = loss.fused_l1_ssim_loss(a, b, lam)
    (L * up).backward()
    return float(L), a.grad.cpu().numpy()


def test_fused_loss_matches_reference_golden():
    d = np.load(os.path.join(G, "loss.npz"))
    L, g = _run(d["img1"], d["img2"])
    assert L == pytest.approx(float(d["loss"]), rel=2e-6)
    np.testing.assert_allclose(g, d["grad"], rtol=2e-4, atol=2e-9)


@pytest.mark.parametrize("shape,lam,up", [((3, 57, 100), 0.2, 1.0), ((3, 16, 16), 0.5, -2.0), ((1, 5, 7), 0.2, 1.0), ((3, 96, 33), 1.0, 3.0)])
def test_fused_loss_matches_numpy_oracle(shape, lam, up):
    rng = np.random.default_rng(sum(shape))
    img, gt = rng.uniform(0, 1, shape).astype(np.float32), rng.uniform(0, 1, shape).astype(np.float32)
    img[0, :2, :3] = gt[0, :2, :3]                      # exact zeros of (x - y): sign(0) = 0
    L, g = _run(img, gt, lam, up)
    Lr, _, _, gr = ssim_ref.l1_ssim_loss(img, gt, lam)
    assert L == pytest.approx(Lr, rel=3e-6)
    scale = np.abs(gr).max() * abs(up)
    assert np.abs(g - up * gr).max() <= 2e-4 * scale


def test_fused_loss_matches_torch_path_at_1080p_and_trains():
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 1, (3, 1080, 1920)).astype(np.float32); gt = rng.uniform(0, 1, (3, 1080, 1920)).astype(np.float32)
    a = torch.tensor(img, device="cuda", requires_grad=True); b = torch.tensor(gt, device="cuda")
    Lt = loss.training_loss(a, b); Lt.backward(); gt_grad = a.grad.clone(); a.grad = None
    Lf = loss.fused_l1_ssim_loss(a, b); Lf.backward()
    assert float(Lf) == pytest.approx(float(Lt), rel=5e-6)
    assert float((a.grad - gt_grad).abs().max()) <= 3e-4 * float(gt_grad.abs().max())
    with pytest.raises(RuntimeError):
        loss.fused_l1_ssim_loss(a.detach().cpu(), b.cpu())
