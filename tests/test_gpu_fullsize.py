"""-m gpu: BASELINE.json's five configurations at full size, every one against the CPU oracle (image AND all gradient
tensors: tests/helpers.parity_report -- radii bit-equal, every pixel over 1e-4 certified by the oracle's decision margin,
per-Gaussian gradient error against the float64 oracle next to the float32 oracle's own), on the scenes SURVEY 8d
prescribes (config 2: table_ds cloud x 17, config 4: tiramisu_ds cloud x 9 seen by 8 ring cameras).
Configs 3 and 5 (1 M @1080p, 5 M @4K) additionally through size-independent properties:
  * sortedness of every tile's slice, ranges partition the list, n_contrib <= slice length;
  * background linearity: image(bg1) - image(bg0) = T_final * (bg1 - bg0);
  * linearity of the backward pass in dL/dimage;
  * invariance under a permutation of the Gaussians, up to exact depth ties: splats with identical
    float32 depth are ordered by index (as upstream's stable sort does), so a vanishing fraction of pixels
    where two such splats overlap may move by ~1e-4.
"""
import numpy as np
import pytest
import torch

from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, synth
from oracle import ref
from tests.helpers import (GRAD_RTOL, assert_image_close, assert_parity, grad_err, hip_forward_backward, oracle_scene,
                           parity_report)

pytestmark = pytest.mark.gpu


def _dump(name, rep):
    """Keeps the report next to the other GPU-run artefacts (gpurun_out/ travels back from the box)."""
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"parity_{name}.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("name,kw", [
    ("cfg1_plumbing_10k_256", {}),                    # 10 k random Gaussians, 256 x 256, SH degree 0
    ("cfg2_table_300k_800", {}),                      # table_ds cloud x 17 = 299 506 Gaussians, 800 x 800
    ("generic_300k_800", {}),                         # round 1's stand-in for config 2 (generic generator)
    ("cfg3_synth_1M_1080p", {}),                      # the headline config
    ("cfg5_stress_5M_4k", {}),                        # 5 M Gaussians, 3840 x 2160
])
def test_config_against_oracle(name, kw):
    sc = synth.make_config(name, **kw)
    rep = parity_report(oracle_scene(sc), sc.dL_dimage)
    _dump(name, rep)
    assert_parity(rep)


@pytest.mark.parametrize("camera", range(8))
def test_config4_tiramisu_ring_camera_against_oracle(camera):
    """Config 4: the tiramisu_ds scene (303 570 Gaussians), each of the 8 ring cameras (1600 x 900) on one GPU."""
    sc = synth.make_config("cfg4_tiramisu_303k_1600x900", camera=camera)
    assert sc.P == 303570 and sc.camera.image_width == 1600 and sc.camera.image_height == 900
    assert abs(sc.camera.tanfovx - 0.6132) < 1e-3
    rep = parity_report(oracle_scene(sc), sc.dL_dimage)
    _dump(f"cfg4_cam{camera}", rep)
    assert rep["num_rendered_reference_rule"] > 1_000_000        # the cloud is in view
    assert_parity(rep)


def _tensors(sc, dev="cuda"):
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float32, device=dev).requires_grad_(g)
    cam = sc.camera
    inp = dict(means3D=t(sc.means3D, True), opacities=t(sc.opacities, True), shs=t(sc.shs, True), scales=t(sc.scales, True),
               rotations=t(sc.rotations, True))
    mk = lambda bg: GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(bg), 1.0,
                                                  t(cam.world_view_transform), t(cam.full_proj_transform), sc.sh_degree,
                                                  t(cam.camera_center), False, False)
    return inp, mk


@pytest.mark.parametrize("name,kw", [("cfg3_synth_1M_1080p", {}), ("cfg5_stress_5M_4k", {})])
def test_fullsize_properties(name, kw):
    from gaussian_transformer_amd import _lib
    from gaussian_transformer_amd.rasterizer import get_backend
    sc = synth.make_config(name, **kw)
    inp, mk = _tensors(sc)
    P, H, W = sc.P, sc.camera.image_height, sc.camera.image_width
    m2 = lambda: torch.zeros((P, 3), device="cuda", requires_grad=True)
    bg0, bg1 = np.array([0.0, 0.0, 0.0], np.float32), np.array([0.9, 0.3, 0.6], np.float32)
    # ---- internal lists ----
    be = get_backend()
    with torch.no_grad():
        e = torch.empty(0, device="cuda")
        n, color0, radii, geom, binning, img = be.forward(mk(bg0), inp["means3D"], inp["shs"], e, inp["opacities"], inp["scales"],
                                                         inp["rotations"], e)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    pl = np.zeros(n, np.uint32); ranges = np.zeros((T, 2), np.uint32)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(be.lib.gsr_debug_read_binning(stream, n, W, H, binning.data_ptr(), img.data_ptr(), None, pl.ctypes.data, ranges.ctypes.data), "read")
    fT = np.zeros((H, W), np.float32); nc = np.zeros((H, W), np.uint32)
    _lib.check(be.lib.gsr_debug_read_image_state(stream, W, H, img.data_ptr(), fT.ctypes.data, nc.ctypes.data), "read")
    depth = np.zeros(P, np.float32)
    _lib.check(be.lib.gsr_debug_read_geom(stream, P, geom.data_ptr(), depth.ctypes.data, None, None, None, None, None), "read")
    lens = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
    nz = lens > 0
    assert int(lens.sum()) == n and (ranges[nz, 0] <= ranges[nz, 1]).all()
    order = np.argsort(ranges[nz, 0], kind="stable")
    starts, ends = ranges[nz][order, 0], ranges[nz][order, 1]
    assert starts[0] == 0 and ends[-1] == n and (starts[1:] == ends[:-1]).all()          # ranges partition the list
    d = depth[pl]
    same_tile = np.ones(n - 1, bool); same_tile[ends[:-1] - 1] = False                  # pairs (j, j+1) inside one tile
    bad = same_tile & ((d[1:] < d[:-1]) | ((d[1:] == d[:-1]) & (pl[1:] < pl[:-1])))
    assert not bad.any()                                                                 # every slice sorted by (depth, id)
    tile_of_pix = (np.arange(H)[:, None] // 16) * ((W + 15) // 16) + (np.arange(W)[None, :] // 16)
    assert (nc <= lens[tile_of_pix]).all()
    assert np.isfinite(color0.cpu().numpy()).all() and (fT >= 0).all() and (fT <= 1).all()
    # ---- background linearity ----
    with torch.no_grad():
        c1, _ = GaussianRasterizer(raster_settings=mk(bg1))(means2D=m2(), **inp)
    diff = (c1 - color0).cpu().numpy()
    np.testing.assert_allclose(diff, fT[None] * (bg1 - bg0)[:, None, None], atol=2e-6)
    # ---- backward linearity in dL ----
    dL = torch.tensor(sc.dL_dimage, device="cuda")
    params = [inp[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")]
    ca, _ = GaussianRasterizer(raster_settings=mk(bg1))(means2D=m2(), **inp)
    ga = torch.autograd.grad(ca, params, grad_outputs=dL)
    cb, _ = GaussianRasterizer(raster_settings=mk(bg1))(means2D=m2(), **inp)
    gb = torch.autograd.grad(cb, params, grad_outputs=-2.5 * dL)
    for x, y in zip(ga, gb):
        x, y = x.cpu().numpy(), y.cpu().numpy()
        assert np.isfinite(x).all()
        assert grad_err(y, -2.5 * x) < GRAD_RTOL      # float atomics: summation order differs run to run
    del ga, gb, ca, cb
    # ---- permutation invariance ----
    perm = torch.tensor(np.random.default_rng(3).permutation(P), device="cuda")
    with torch.no_grad():
        cp, rp = GaussianRasterizer(raster_settings=mk(bg1))(means2D=m2(), **{k: v[perm] for k, v in inp.items()})
    assert torch.equal(rp.cpu(), radii[perm].cpu())
    assert_image_close(cp.cpu().numpy(), c1.cpu().numpy(), atol=2e-6, outlier_frac=1e-3, outlier_max=5e-3)
