"""Pins the CPU restatement (oracle/) and the host helpers against fixtures generated from
the reference's own importable modules (oracle/make_golden.py -> tests/golden/*.npz)."""
import math
import os

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import camera, loss
from oracle import ref

G = os.path.join(os.path.dirname(__file__), "golden")


def _ident_cam(W=64, H=64):
    from gaussian_transformer_amd.synth import identity_camera
    return identity_camera(W, H)


@pytest.mark.parametrize("prec,tol", [("f64", 1e-12), ("f32", 2e-6)])
@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_matches_reference_eval_sh(prec, tol, deg):
    """S6 vs utils/sh_utils.py:eval_sh + clamp_min(.+0.5, 0) (gaussian_renderer/__init__.py:78)."""
    d = np.load(os.path.join(G, "sh_eval.npz"))
    dirs, sh_view = d["dirs"], d["sh_view"]                      # [64,3], [64,3,16]
    campos = np.array([0.0, 0.0, 5.0])
    means = campos[None] + dirs                                   # direction from campos == dirs
    cam = _ident_cam()
    shs = np.transpose(sh_view, (0, 2, 1))                        # storage layout [P,M,3]
    S = ref.Scene(W=64, H=64, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform,
                  projmatrix=cam.full_proj_transform, campos=campos, means3D=means,
                  opacities=np.full((64,), 0.5), sh_degree=deg, shs=shs,
                  scales=np.full((64, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (64, 1)))
    f = ref.get(prec).forward(S)
    assert (f["radii"] > 0).all()
    geom = f["state"].geom()
    np.testing.assert_allclose(geom["rgb"], d[f"rgb_clamped_deg{deg}"], atol=tol, rtol=0)
    np.testing.assert_array_equal(geom["clamped"].astype(bool), (d[f"rgb_raw_deg{deg}"] + 0.5) < 0)


@pytest.mark.parametrize("prec,tol", [("f64", 2e-7), ("f32", 2e-6)])
@pytest.mark.parametrize("mod", [1.0, 0.5])
def test_cov3d_matches_reference_build_scaling_rotation(prec, tol, mod):
    """S2 vs utils/general_utils.py:64-110 + scene/gaussian_model.py:27-31 (fixture is float32)."""
    d = np.load(os.path.join(G, "cov3d.npz"))
    P = d["scales"].shape[0]
    cam = _ident_cam()
    means = np.stack([np.zeros(P), np.zeros(P), np.full(P, 4.0)], 1)
    S = ref.Scene(W=64, H=64, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, viewmatrix=cam.world_view_transform,
                  projmatrix=cam.full_proj_transform, campos=cam.camera_center, means3D=means,
                  opacities=np.full((P,), 0.5), colors_precomp=np.full((P, 3), 0.5),
                  scales=d["scales"], rotations=d["quats_unit"], scale_modifier=mod)
    f = ref.get(prec).forward(S)
    cov = f["state"].geom()["cov3D"]
    ref_cov = d[f"cov6_mod{mod}"]
    scale = np.abs(ref_cov).max()
    assert np.abs(cov - ref_cov).max() <= tol * max(scale, 1.0)


def test_camera_matrices_match_reference():
    d = np.load(os.path.join(G, "camera.npz"))
    # known-answer anchors of SURVEY.md 8c
    assert camera.focal2fov(3049.779011853469, 4032) == pytest.approx(1.1681823647483933, abs=1e-15)
    assert camera.focal2fov(3049.779011853469, 2268) == pytest.approx(0.7119775858360171, abs=1e-15)
    assert math.tan(0.5 * float(d["table_fovx"])) == pytest.approx(0.6610315016807722, abs=1e-12)
    assert math.tan(0.5 * float(d["table_fovy"])) == pytest.approx(0.37183021969543434, abs=1e-12)
    for name in ("table", "tiramisu"):
        fx, fy = float(d[f"{name}_fovx"]), float(d[f"{name}_fovy"])
        np.testing.assert_array_equal(camera.projection_matrix(0.01, 100.0, fx, fy), d[f"{name}_proj"])
        assert camera.fov2focal(fx, 4032) == pytest.approx(float(d[f"{name}_focal_back"]), rel=1e-15)
    np.testing.assert_array_equal(camera.world_to_view(np.eye(3), np.array([0.0, 0.0, 3.0])), d["w2v_identity"])
    fovx, fovy = float(d["table_fovx"]), float(d["table_fovy"])
    for i in range(4):
        m = camera.world_to_view(d["R"][i], d["T"][i], d["translate"][i], float(d["scale"][i]))
        np.testing.assert_array_equal(m, d["w2v"][i])
        c = camera.make_camera(d["R"][i], d["T"][i], fovx, fovy, 4032, 2268, d["translate"][i], float(d["scale"][i]))
        np.testing.assert_array_equal(c.world_view_transform, d["world_view_transform"][i])
        np.testing.assert_allclose(c.full_proj_transform, d["full_proj_transform"][i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(c.camera_center, d["camera_center"][i], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("prec,tol", [("f64", 1e-5), ("f32", 1e-4)])
def test_projection_epsilon_and_layout_match_geom_transform_points(prec, tol):
    """S1/S5: pixel centre from NDC with the +1e-7 divide (utils/graphics_utils.py:22-29)."""
    d = np.load(os.path.join(G, "geom_transform.npz"))
    c = np.load(os.path.join(G, "camera.npz"))
    pts, full, out = d["points"], d["matrix"], d["out"]
    wvt = c["world_view_transform"][0]
    W, H = 640, 360
    tanx, tany = math.tan(0.5 * float(c["table_fovx"])), math.tan(0.5 * float(c["table_fovy"]))
    P = pts.shape[0]
    S = ref.Scene(W=W, H=H, tanfovx=tanx, tanfovy=tany, viewmatrix=wvt, projmatrix=full, campos=c["camera_center"][0],
                  means3D=pts, opacities=np.full((P,), 0.5), colors_precomp=np.full((P, 3), 0.5),
                  scales=np.full((P, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (P, 1)))
    f = ref.get(prec).forward(S)
    vis = f["radii"] > 0
    assert vis.sum() >= 5
    xy = f["state"].geom()["xy"][vis]
    ndc_x = (2.0 * xy[:, 0] + 1.0) / W - 1.0
    ndc_y = (2.0 * xy[:, 1] + 1.0) / H - 1.0
    np.testing.assert_allclose(ndc_x, out[vis, 0], atol=tol, rtol=tol)
    np.testing.assert_allclose(ndc_y, out[vis, 1], atol=tol, rtol=tol)
    # depth = view-space z with the transposed-row-major layout
    pv = np.concatenate([pts, np.ones((P, 1), np.float32)], 1) @ wvt
    np.testing.assert_allclose(f["state"].geom()["depth"][vis], pv[vis, 2], rtol=1e-5)
    np.testing.assert_array_equal(ref.get(prec).mark_visible(pts, wvt), pv[:, 2] > 0.2)


def test_loss_matches_reference_l1_ssim_psnr():
    d = np.load(os.path.join(G, "loss.npz"))
    a = torch.tensor(d["img1"], requires_grad=True); b = torch.tensor(d["img2"])
    assert loss.l1_loss(a, b).item() == pytest.approx(float(d["l1"]), rel=1e-6)
    assert loss.ssim(a, b).item() == pytest.approx(float(d["ssim"]), rel=1e-5)
    L = loss.training_loss(a, b)
    assert L.item() == pytest.approx(float(d["loss"]), rel=1e-6)
    L.backward()
    np.testing.assert_allclose(a.grad.numpy(), d["grad"], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(loss.psnr(a.detach()[None], b[None]).numpy(), d["psnr"], rtol=1e-6)


def test_numpy_loss_oracle_matches_reference_values_and_gradient():
    """oracle/ssim_ref.py (the checker of the fused HIP loss) vs the reference's own l1/ssim/loss and autograd."""
    from oracle import ssim_ref
    d = np.load(os.path.join(G, "loss.npz"))
    loss_v, l1, ssim, grad = ssim_ref.l1_ssim_loss(d["img1"], d["img2"], 0.2)
    assert l1 == pytest.approx(float(d["l1"]), rel=1e-6)
    assert ssim == pytest.approx(float(d["ssim"]), rel=1e-5)
    assert loss_v == pytest.approx(float(d["loss"]), rel=1e-6)
    np.testing.assert_allclose(grad, d["grad"], rtol=2e-4, atol=2e-9)
