"""-m gpu: distCUDA2 equivalent (include/gsr_knn.h) vs scipy's exact k-d tree.
"Parity unpinned": simple_knn is an empty submodule in the reference; the semantics come from its call site
(scene/gaussian_model.py:134) -- mean squared distance to the 3 nearest other points."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu


def _oracle(p):
    d, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(axis=1)


@pytest.mark.parametrize("kind,n", [("uniform", 20000), ("clustered", 50000), ("planar", 5000), ("tiny", 5), ("sfm_like", 300000)])
def test_distCUDA2_matches_kdtree(kind, n):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(len(kind) + n)
    if kind == "uniform":
        p = rng.uniform(-1, 1, (n, 3))
    elif kind == "clustered":
        c = rng.normal(0, 5, (50, 3)); p = c[rng.integers(0, 50, n)] + rng.normal(0, 0.05, (n, 3)); p[:100] = rng.normal(0, 200, (100, 3))
    elif kind == "planar":
        p = np.concatenate([rng.uniform(-1, 1, (n, 2)), np.zeros((n, 1))], 1); p[:10] = p[10:20]          # duplicates
    elif kind == "tiny":
        p = rng.normal(size=(n, 3))
    else:
        p = rng.normal(0, 1, (n, 3)) * np.array([5.0, 5.0, 1.0]) + np.array([0.4, 1.0, 6.2])
    p = p.astype(np.float32)
    got = distCUDA2(torch.tensor(p, device="cuda")).cpu().numpy()
    ref = _oracle(p)
    scale = np.maximum(ref, 1e-12)
    assert np.abs(got - ref).max() <= 1e-4 * max(ref.max(), 1e-12) or (np.abs(got - ref) / scale).max() < 1e-3
    assert (np.abs(got - ref) / np.maximum(ref, 1e-9)).max() < 2e-3 or np.abs(got - ref).max() < 1e-9


def test_distCUDA2_drop_in_use_as_in_create_from_pcd():
    """scene/gaussian_model.py:134-135: scales = log(sqrt(clamp_min(distCUDA2(points), 1e-7)))"""
    from simple_knn._C import distCUDA2
    pts = torch.tensor(np.random.default_rng(0).normal(size=(1000, 3)).astype(np.float32), device="cuda")
    dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    assert scales.shape == (1000, 3) and torch.isfinite(scales).all()
    with pytest.raises(RuntimeError):
        distCUDA2(pts.cpu())
