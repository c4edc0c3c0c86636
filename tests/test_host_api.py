"""CPU tests of the host side: C-ABI library loads and exports every symbol of include/gsr.h,
the product path fails loudly without a GPU, and the Python API's validation / autograd plumbing
(exercised through an oracle-backed stand-in backend -- tests/oracle_backend.py)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, rasterizer, synth
from gaussian_transformer_amd.model import GaussianParams
from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render
from tests.oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    from gaussian_transformer_amd.build import build_hip
    path = build_hip()                      # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "gsr.h")).read() + open(os.path.join(ROOT, "include", "gsr_loss.h")).read() + open(os.path.join(ROOT, "include", "gsr_knn.h")).read() + open(os.path.join(ROOT, "include", "gsr_optim.h")).read()
    declared = set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", header)) - {"gsr_alloc_fn"}
    assert {"gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_workspace_sizes", "gsr_binning_bytes",
            "gsr_last_error", "gsr_abi_version"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_lib.SIGNATURES) == declared            # the ctypes stub binds exactly the header
    assert _lib.load().gsr_abi_version() == 2


def test_drop_in_module_name_and_settings_fields():
    import diff_gaussian_rasterization as d
    assert d.GaussianRasterizer is GaussianRasterizer and d.GaussianRasterizationSettings is GaussianRasterizationSettings
    # field order / names of gaussian_renderer/__init__.py:36-49
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def _settings(cam, bg, deg):
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
    return GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(bg), 1.0,
                                         t(cam.world_view_transform), t(cam.full_proj_transform), deg, t(cam.camera_center),
                                         False, False)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = synth.make_scene(P=10, width=32, height=32, sh_degree=0, s0=0.1)
    prev = rasterizer._set_backend_for_tests(None)
    try:
        rast = GaussianRasterizer(raster_settings=_settings(sc.camera, sc.bg, 0))
        t = lambda a: torch.tensor(a)
        with pytest.raises(RuntimeError, match="HIP device|no CPU fallback"):
            rast(means3D=t(sc.means3D), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), shs=t(sc.shs),
                 scales=t(sc.scales), rotations=t(sc.rotations))
    finally:
        rasterizer._set_backend_for_tests(prev)


@pytest.fixture
def oracle_backend():
    prev = rasterizer._set_backend_for_tests(OracleBackend())
    yield
    rasterizer._set_backend_for_tests(prev)


def test_validation_messages(oracle_backend):
    sc = synth.make_scene(P=10, width=32, height=32, sh_degree=0, s0=0.1)
    rast = GaussianRasterizer(raster_settings=_settings(sc.camera, sc.bg, 0))
    t = lambda a: torch.tensor(a)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=t(sc.means3D), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), scales=t(sc.scales), rotations=t(sc.rotations))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=t(sc.means3D), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), shs=t(sc.shs),
             colors_precomp=torch.zeros(10, 3), scales=t(sc.scales), rotations=t(sc.rotations))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=t(sc.means3D), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), shs=t(sc.shs))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        rast(means3D=torch.zeros(10, 4), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), shs=t(sc.shs),
             scales=t(sc.scales), rotations=t(sc.rotations))
    with pytest.raises(RuntimeError, match="float32"):
        rast(means3D=t(sc.means3D).double(), means2D=torch.zeros(10, 3), opacities=t(sc.opacities), shs=t(sc.shs),
             scales=t(sc.scales), rotations=t(sc.rotations))


def test_render_wrapper_dict_and_means2d_gradient(oracle_backend):
    """render() restates gaussian_renderer/__init__.py:18-100: dict keys, retain_grad on the dummy
    screenspace tensor, densification statistic of scene/gaussian_model.py:405-407."""
    sc = synth.make_scene(P=300, width=48, height=40, sh_degree=2, s0=0.08, seed=3)
    pc = GaussianParams.from_synthetic(sc, "cpu")
    cam = TorchCamera(sc.camera, "cpu")
    pkg = render(cam, pc, PipelineParams(), torch.tensor(sc.bg))
    assert set(pkg) == {"render", "viewspace_points", "visibility_filter", "radii"}
    assert pkg["render"].shape == (3, 40, 48) and pkg["radii"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool
    (pkg["render"] * torch.tensor(sc.dL_dimage)).sum().backward()
    vsp = pkg["viewspace_points"]
    assert vsp.grad is not None and vsp.grad.shape == (300, 3) and float(vsp.grad[:, 2].abs().max()) == 0.0
    for p in pc.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    pc.add_densification_stats(vsp, pkg["visibility_filter"])
    assert float(pc.denom.sum()) == float(pkg["visibility_filter"].sum())
    # the two optional Python paths of the wrapper give the same image
    img0 = pkg["render"].detach()
    img1 = render(cam, pc, PipelineParams(convert_SHs_python=True), torch.tensor(sc.bg))["render"].detach()
    img2 = render(cam, pc, PipelineParams(compute_cov3D_python=True), torch.tensor(sc.bg))["render"].detach()
    assert float((img0 - img1).abs().max()) < 1e-5 and float((img0 - img2).abs().max()) < 1e-5
    with torch.no_grad():
        img3 = render(cam, pc, PipelineParams(), torch.tensor(sc.bg), scaling_modifier=0.5)["render"]
    assert not img3.requires_grad and float((img3 - img0).abs().max()) > 1e-3


def test_gradient_arena_views(oracle_backend):
    """Backward can write the parameter gradients straight into one flat bucket (no packing copy)."""
    from gaussian_transformer_amd.rasterizer import arena_floats, gradient_arena
    sc = synth.make_scene(P=50, width=32, height=32, sh_degree=1, s0=0.1, max_sh_degree=1)
    t = lambda a: torch.tensor(a).requires_grad_(True)
    inp = dict(means3D=t(sc.means3D), shs=t(sc.shs), opacities=t(sc.opacities), scales=t(sc.scales), rotations=t(sc.rotations))
    rast = GaussianRasterizer(raster_settings=_settings(sc.camera, sc.bg, 1))
    dL = torch.tensor(sc.dL_dimage)
    c, _ = rast(means2D=torch.zeros(50, 3, requires_grad=True), **inp)
    ref_g = torch.autograd.grad(c, list(inp.values()), grad_outputs=dL)
    # the stand-in backend does not use the arena itself; check the carving helper and size rule
    assert arena_floats(50, 4) == 50 * (3 + 12 + 1 + 3 + 4)
    flat = torch.zeros(arena_floats(50, 4))
    with gradient_arena(flat) as f:
        assert f is flat and rasterizer._grad_arena is flat
    assert rasterizer._grad_arena is None
    assert all(torch.isfinite(g).all() for g in ref_g)


def test_every_option_the_library_accepts_is_documented_in_the_header():
    """include/gsr.h is the contract: a knob gsr_set_option / gsr_get_option knows must be described there."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "gaussian_transformer_amd", "csrc", "gsr_api.hip")).read()
    hdr = open(os.path.join(root, "include", "gsr.h")).read()
    names = sorted(set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', api)))
    assert len(names) > 15
    missing = [n for n in names if f'"{n}"' not in hdr]
    assert not missing, missing
