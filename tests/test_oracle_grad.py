"""The analytic backward of oracle/gsr_ref.c vs autograd of the float64 dense restatement
(oracle/dense_ref.py) and vs central finite differences of the oracle's own forward."""
import numpy as np
import pytest
import torch

from gaussian_transformer_amd import synth
from oracle import dense_ref, ref
from tests.helpers import grad_err, oracle_scene


def _dense(sc, S, dL, use_cov=False, use_colors=False):
    cam = sc.camera
    t = lambda a, rg=True: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=rg)
    P = sc.P
    inp = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), means2D=torch.zeros(P, 3, dtype=torch.float64, requires_grad=True))
    if use_colors:
        inp["colors_precomp"] = t(S.colors_precomp)
    else:
        inp["shs"] = t(sc.shs)
    if use_cov:
        inp["cov3D_precomp"] = t(S.cov3D_precomp)
    else:
        inp["scales"] = t(sc.scales); inp["rotations"] = t(sc.rotations)
    col, radii, aux = dense_ref.dense_render(
        S.W, S.H, S.tanfovx, S.tanfovy, t(S.viewmatrix, False), t(S.projmatrix, False), t(S.campos, False), t(S.bg, False),
        sh_degree=S.sh_degree, scale_modifier=S.scale_modifier, **inp)
    (col * torch.tensor(dL)).sum().backward()
    return col.detach().numpy(), radii.numpy(), {k: v.grad.numpy() for k, v in inp.items()}


CASES = [
    dict(P=48, width=40, height=36, sh_degree=3, s0=0.08, seed=0, zmin=0.1, zmax=6.0, bg=(0.3, 0.1, 0.7)),
    dict(P=64, width=33, height=47, sh_degree=1, s0=0.3, seed=3, bg=(0.0, 0.0, 0.0)),
    dict(P=30, width=16, height=16, sh_degree=0, s0=1.0, seed=4, bg=(1.0, 1.0, 1.0)),
    dict(P=40, width=50, height=20, sh_degree=2, s0=0.05, seed=7, tanfovx=0.05, bg=(0.2, 0.2, 0.2)),  # narrow FoV: 1.3*tanfov clamp active
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_f64_oracle_matches_dense_autograd(case):
    kw = dict(CASES[case])
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc, scale_modifier=1.3)
    dL = np.random.default_rng(5).normal(size=(3, S.H, S.W))
    r = ref.get("f64")
    f = r.forward(S); g = r.backward(f, dL)
    col, radii, dg = _dense(sc, S, dL)
    np.testing.assert_array_equal(radii, f["radii"])
    assert np.abs(col - f["color"]).max() < 1e-12
    # 2e-5: the dense form has no 1e-7 in 1/(det^2+1e-7) (dense_ref.py header)
    for name, key in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"),
                      ("scales", "dL_dscales"), ("rotations", "dL_drots")]:
        assert grad_err(g[key], dg[name]) < 2e-5, name
    assert grad_err(g["dL_dopacity"], dg["opacities"].reshape(-1)) < 2e-5


def test_f64_oracle_precomputed_inputs_match_dense_autograd():
    sc = synth.make_scene(P=40, width=32, height=32, sh_degree=0, s0=0.2, seed=11)
    rng = np.random.default_rng(2)
    A = rng.normal(size=(sc.P, 3, 3)) * 0.2
    cov = A @ np.transpose(A, (0, 2, 1)) + 0.01 * np.eye(3)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    colors = rng.uniform(0, 1, size=(sc.P, 3))
    S = oracle_scene(sc, shs=None, colors_precomp=colors, scales=None, rotations=None, cov3D_precomp=cov6)
    dL = rng.normal(size=(3, S.H, S.W))
    r = ref.get("f64")
    f = r.forward(S); g = r.backward(f, dL)
    col, radii, dg = _dense(sc, S, dL, use_cov=True, use_colors=True)
    assert np.abs(col - f["color"]).max() < 1e-12
    assert grad_err(g["dL_dcolors"], dg["colors_precomp"]) < 1e-10
    assert grad_err(g["dL_dcov3D"], dg["cov3D_precomp"]) < 2e-5
    assert grad_err(g["dL_dmeans3D"], dg["means3D"]) < 2e-5


def test_f64_oracle_backward_matches_finite_differences():
    """Independent of the dense restatement: central differences on the oracle's own forward."""
    sc = synth.make_scene(P=12, width=24, height=24, sh_degree=1, s0=0.25, seed=21, bg=(0.1, 0.2, 0.3))
    S = oracle_scene(sc)
    r = ref.get("f64")
    dL = np.random.default_rng(1).normal(size=(3, S.H, S.W))
    f = r.forward(S); g = r.backward(f, dL)
    fields = [("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
              ("opacities", "dL_dopacity"), ("shs", "dL_dsh")]
    rng = np.random.default_rng(0)
    h = 1e-6
    for name, key in fields:
        base = np.asarray(getattr(S, name), dtype=np.float64)
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in base.shape)
            vals = []
            for sgn in (+1, -1):
                pert = base.copy(); pert[idx] += sgn * h
                S2 = oracle_scene(sc, **{name: pert})
                vals.append((r.forward(S2)["color"] * dL).sum())
            fd = (vals[0] - vals[1]) / (2 * h)
            an = np.asarray(g[key]).reshape(base.shape)[idx]
            assert abs(fd - an) <= 1e-4 * max(1.0, abs(an)) + 1e-6, (name, idx, fd, an)


def test_f32_oracle_tracks_f64_oracle():
    sc = synth.make_scene(P=400, width=96, height=64, sh_degree=3, s0=0.05, seed=9)
    S = oracle_scene(sc)
    dL = np.random.default_rng(3).normal(size=(3, S.H, S.W)) / (3 * S.H * S.W)
    r32, r64 = ref.get("f32"), ref.get("f64")
    f32, f64 = r32.forward(S), r64.forward(S)
    np.testing.assert_array_equal(f32["radii"], f64["radii"])
    assert np.abs(f32["color"] - f64["color"]).max() < 1e-5
    g32, g64 = r32.backward(f32, dL), r64.backward(f64, dL)
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drots"):
        assert grad_err(g32[k], g64[k]) < 1e-4, k


def test_oracle_edge_cases():
    r = ref.get("f32")
    cam = synth.identity_camera(32, 20)
    base = dict(W=32, H=20, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, campos=cam.camera_center, bg=np.array([0.2, 0.4, 0.6]))
    # P = 0: image of zeros, NOT background (SURVEY.md 8a edge cases)
    f = r.forward(ref.Scene(means3D=np.zeros((0, 3)), opacities=np.zeros((0,)), colors_precomp=np.zeros((0, 3)),
                            cov3D_precomp=np.zeros((0, 6)), **base))
    assert f["num_rendered"] == 0 and np.all(f["color"] == 0)
    # everything behind the near plane: background everywhere, radii 0
    f = r.forward(ref.Scene(means3D=np.array([[0, 0, 0.1], [0, 0, -3.0]]), opacities=np.array([0.9, 0.9]),
                            colors_precomp=np.ones((2, 3)), scales=np.full((2, 3), 0.1),
                            rotations=np.tile([1.0, 0, 0, 0], (2, 1)), **base))
    assert (f["radii"] == 0).all() and f["num_rendered"] == 0
    np.testing.assert_allclose(f["color"][:, 3, 5], [0.2, 0.4, 0.6], atol=1e-7)
    # one opaque splat in the centre: W,H not multiples of 16, keys sorted, ranges consistent
    f = r.forward(ref.Scene(means3D=np.array([[0, 0, 2.0], [0.05, 0, 1.5]]), opacities=np.array([0.9, 0.7]),
                            colors_precomp=np.array([[1.0, 0, 0], [0, 1.0, 0]]), scales=np.full((2, 3), 0.2),
                            rotations=np.tile([1.0, 0, 0, 0], (2, 1)), **base))
    b = f["state"].binning()
    assert (np.diff(b["keys"].astype(np.uint64)) >= 0).all() if len(b["keys"]) > 1 else True
    assert f["num_rendered"] == len(b["keys"]) == int(f["state"].geom()["tiles_touched"].sum())
    centre = f["color"][:, 10, 16]
    assert centre[1] > centre[0] > 0.0      # nearer green splat dominates the centre pixel
