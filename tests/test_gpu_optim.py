"""-m gpu: HipAdam (include/gsr_optim.h) against torch.optim.Adam on the reference's optimiser layout."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5003, 3), (5003, 1, 3), (5003, 15, 3), (5003, 1), (5003, 3), (5003, 4), (7,)]     # 7: unaligned tail / tiny group
    return [torch.randn(s, generator=g).to(dev).requires_grad_(True) for s in shapes]


def test_hip_adam_matches_torch_adam_over_steps():
    from gaussian_transformer_amd.optim import HipAdam
    dev = torch.device("cuda", 0)
    lrs = [0.00016, 0.0025, 0.000125, 0.05, 0.005, 0.001, 0.01]
    a, b = _params(0, dev), _params(0, dev)
    oa = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=1e-15)
    ob = HipAdam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(1)
    for step in range(12):
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (10.0 ** (-(step % 4)))).to(dev)
            if step == 5:
                gr[::3] = 0.0                                  # exact zeros: denominators at eps scale
            pa.grad = gr.clone(); pb.grad = gr.clone()
        oa.step(); ob.step()
        if step == 7:                                          # a learning-rate change mid-run (update_learning_rate)
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 0.00005
    for pa, pb in zip(a, b):
        sa, sb = oa.state[pa], ob.state[pb]
        assert int(sa["step"]) == int(sb["step"]) == 12
        # same formulas, different rounding (fma contraction): a few ulp of the tensor's scale
        close = lambda x, y, tol: float((x - y).abs().max()) <= tol * float(y.abs().max()) + 1e-30
        assert close(sb["exp_avg"], sa["exp_avg"], 2e-6), float((sb["exp_avg"] - sa["exp_avg"]).abs().max())
        assert close(sb["exp_avg_sq"], sa["exp_avg_sq"], 1e-5), float((sb["exp_avg_sq"] - sa["exp_avg_sq"]).abs().max() / sa["exp_avg_sq"].abs().max())
        assert close(pb, pa, 2e-5), float((pb - pa).abs().max())


def test_hip_adam_under_density_control():
    """The controller's state surgery works on HipAdam as on torch.optim.Adam (same keys, same shapes)."""
    from gaussian_transformer_amd import synth
    from gaussian_transformer_amd.densify import DensityController
    from gaussian_transformer_amd.model import GaussianParams
    dev = torch.device("cuda", 0)
    sc = synth.make_scene(P=500, width=64, height=64, sh_degree=1, s0=0.05, seed=3)
    m = GaussianParams.from_synthetic(sc, dev)
    ctl = DensityController(m, adam="hip")
    for g in ctl.optimizer.param_groups:
        g["params"][0].grad = torch.ones_like(g["params"][0]) * 1e-3
    ctl.optimizer.step()
    with torch.no_grad():
        m.xyz_gradient_accum += 1.0; m.denom += 1.0
        n = ctl.densify_and_prune(0.0002, 0.005, 3.0, None, generator=torch.Generator(device=dev).manual_seed(0))
    assert n["cloned"] + n["split"] > 0
    P = m._xyz.shape[0]
    for g in ctl.optimizer.param_groups:
        p = g["params"][0]
        assert p.shape[0] == P and ctl.optimizer.state[p]["exp_avg"].shape == p.shape
        p.grad = torch.ones_like(p) * 1e-3
    before = m._xyz.detach().clone()
    ctl.optimizer.step()
    assert not torch.equal(before, m._xyz.detach())
