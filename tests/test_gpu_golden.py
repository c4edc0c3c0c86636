"""-m gpu: the fixtures generated from the reference's own Python modules (tests/golden/*.npz, oracle/make_golden.py) fed
to the HIP kernels directly -- not to the oracle -- and read back through gsr_debug_read_geom:
  sh_eval.npz         utils/sh_utils.py eval_sh + the +0.5 / clamp of gaussian_renderer/__init__.py:78  -> preprocess S6
  cov3d.npz           utils/general_utils.py build_scaling_rotation / strip_symmetric                  -> preprocess S2
  geom_transform.npz  utils/graphics_utils.py geom_transform_points (the 1e-7 perspective epsilon)      -> preprocess S1/S5
  camera.npz          the matrices those points are projected with (scene/cameras.py:54-57 layout)
"""
import os

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import _lib
from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, get_backend

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def _t(a):
    return torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=DEV) if a is not None else torch.empty(0, device=DEV)


def _forward_geom(P, W, H, tanfovx, tanfovy, view, proj, campos, means, opac, deg=0, shs=None, colors=None, scales=None, rots=None,
                  cov=None, mod=1.0):
    be = get_backend()
    rs = GaussianRasterizationSettings(H, W, tanfovx, tanfovy, _t(np.zeros(3)), mod, _t(view), _t(proj), deg, _t(campos), False, False)
    n, color, radii, geom, binning, img = be.forward(rs, _t(means), _t(shs), _t(colors), _t(opac), _t(scales), _t(rots), _t(cov))
    d = dict(depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32), conic_o=np.zeros((P, 4), np.float32),
             rgb=np.zeros((P, 3), np.float32), tiles=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8))
    _lib.check(be.lib.gsr_debug_read_geom(torch.cuda.current_stream().cuda_stream, P, geom.data_ptr(),
                                          *[d[k].ctypes.data for k in ("depth", "xy", "conic_o", "rgb", "tiles", "clamped")]), "read geom")
    d["radii"] = radii.cpu().numpy(); d["color"] = color.cpu().numpy()
    return d


def _identity_cam(W, H, tanfovx):
    from gaussian_transformer_amd.synth import identity_camera
    c = identity_camera(W, H, tanfovx)
    return c, c.world_view_transform, c.full_proj_transform


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colour_stage_reproduces_the_reference_eval_sh(deg):
    g = np.load(os.path.join(G, "sh_eval.npz"))
    dirs, sh_view = g["dirs"], g["sh_view"]                       # [64,3], [64,3,16] (the reference's eval_sh layout)
    P = dirs.shape[0]
    cam, view, proj = _identity_cam(256, 256, 1.0)
    campos = np.array([0.0, 0.0, 5.0], np.float32)               # the SH direction is normalize(p - campos): put every
    means = (campos[None] + dirs).astype(np.float32)             # Gaussian one unit from campos along its fixture direction
    shs = np.ascontiguousarray(np.transpose(sh_view, (0, 2, 1))).astype(np.float32)     # storage layout [P, M, 3]
    d = _forward_geom(P, 256, 256, cam.tanfovx, cam.tanfovy, view, proj, campos, means, np.full((P, 1), 0.5, np.float32), deg=deg, shs=shs,
                      scales=np.full((P, 3), 0.01, np.float32), rots=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))
    assert (d["radii"] > 0).all()
    want = g[f"rgb_clamped_deg{deg}"]
    # the direction is re-derived from float32 positions 5 units from the origin: ~1e-6 relative in the direction
    np.testing.assert_allclose(d["rgb"], want, rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(d["clamped"].astype(bool), (g[f"rgb_raw_deg{deg}"] + 0.5) < 0)       # the clamp mask the backward pass uses


@pytest.mark.parametrize("mod", [1.0, 0.5])
def test_cov3d_stage_reproduces_the_reference_covariance(mod):
    """Scales + (caller-normalised) quaternions through the kernel's own S2 must project exactly like the reference's packed
    covariance fed through cov3D_precomp: conic, radius, tile count and rendered image."""
    g = np.load(os.path.join(G, "cov3d.npz"))
    P = g["scales"].shape[0]
    rng = np.random.default_rng(0)
    cam, view, proj = _identity_cam(320, 240, 0.7)
    z = rng.uniform(2.0, 6.0, P)
    means = np.stack([rng.uniform(-0.5, 0.5, P) * z, rng.uniform(-0.4, 0.4, P) * z, z], 1).astype(np.float32)
    opac = rng.uniform(0.2, 0.9, (P, 1)).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    kw = dict(P=P, W=320, H=240, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, view=view, proj=proj, campos=np.zeros(3), means=means, opac=opac,
              colors=cols)
    a = _forward_geom(scales=g["scales"], rots=g["quats_unit"], mod=mod, **kw)
    b = _forward_geom(cov=g[f"cov6_mod{mod}"], **kw)                   # mod already folded in by the reference (scaling_modifier * s)
    assert (a["radii"] > 0).sum() > 50
    np.testing.assert_array_equal(a["radii"], b["radii"])
    np.testing.assert_array_equal(a["tiles"], b["tiles"])
    np.testing.assert_allclose(a["conic_o"], b["conic_o"], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(a["color"], b["color"], atol=2e-5)


def test_projection_stage_reproduces_geom_transform_points():
    g = np.load(os.path.join(G, "geom_transform.npz"))
    c = np.load(os.path.join(G, "camera.npz"))
    pts, full, ndc = g["points"], g["matrix"], g["out"]
    view = c["world_view_transform"][0]
    np.testing.assert_array_equal(full, c["full_proj_transform"][0])
    P, W, H = pts.shape[0], 640, 360
    tanx, tany = float(np.tan(c["table_fovx"] / 2)), float(np.tan(c["table_fovy"] / 2))
    d = _forward_geom(P, W, H, tanx, tany, view, full, c["camera_center"][0], pts, np.full((P, 1), 0.5, np.float32),
                      colors=np.full((P, 3), 0.5, np.float32), scales=np.full((P, 3), 0.01, np.float32),
                      rots=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))
    zview = (np.concatenate([pts, np.ones((P, 1), np.float32)], 1) @ view)[:, 2]
    vis = zview > 0.2
    assert vis.sum() >= 50 and (~vis).sum() >= 4                   # the fixture has points behind / too near the camera
    np.testing.assert_array_equal(d["radii"][~vis], 0)
    assert (d["radii"][vis] > 0).all()
    want = np.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)     # S5 on the reference's NDC
    np.testing.assert_allclose(d["xy"][vis], want[vis], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(d["depth"][vis], zview[vis], rtol=1e-5)
