"""-m gpu: render()'s optional branches on the HIP device (gaussian_renderer/__init__.py:59-82 of the reference):
pipe.convert_SHs_python (colours evaluated by torch, handed over as colors_precomp), pipe.compute_cov3D_python (covariance
built by torch, handed over as cov3D_precomp) and override_color -- each against the default path and against the oracle,
images and the gradients of the model's raw parameters."""
import numpy as np
import pytest
import torch

from gaussian_transformer_amd import synth
from gaussian_transformer_amd.model import GaussianParams
from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render
from oracle import ref
from tests.helpers import GRAD_RTOL, assert_image_close, grad_err, oracle_scene

pytestmark = pytest.mark.gpu


def _run(sc, pipe, override=None):
    pc = GaussianParams.from_synthetic(sc, "cuda")
    cam = TorchCamera(sc.camera, "cuda")
    bg = torch.tensor(sc.bg, device="cuda")
    oc = None if override is None else torch.tensor(override, device="cuda", requires_grad=True)
    pkg = render(cam, pc, pipe, bg, override_color=oc)
    dL = torch.tensor(sc.dL_dimage, device="cuda") * (3.0 * sc.camera.image_height * sc.camera.image_width)
    (pkg["render"] * dL).sum().backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), pc.parameters())
             if p.grad is not None}
    grads["means2D"] = pkg["viewspace_points"].grad.detach().cpu().numpy()
    if oc is not None:
        grads["override"] = oc.grad.detach().cpu().numpy()
    return pkg["render"].detach().cpu().numpy(), pkg["radii"].cpu().numpy(), grads


@pytest.mark.parametrize("deg", [1, 3])
def test_python_sh_and_python_covariance_branches_match_the_default_path(deg):
    sc = synth.make_scene(P=3000, width=112, height=80, sh_degree=deg, s0=0.04, seed=31, bg=(0.2, 0.1, 0.4), max_sh_degree=3)
    img0, rad0, g0 = _run(sc, PipelineParams())
    # the default path itself against the oracle (same scene, activated parameters)
    S = oracle_scene(sc)
    f = ref.get("f32").forward(S)
    np.testing.assert_array_equal(rad0, f["radii"])
    assert_image_close(img0, f["color"])
    for pipe in (PipelineParams(convert_SHs_python=True), PipelineParams(compute_cov3D_python=True),
                 PipelineParams(convert_SHs_python=True, compute_cov3D_python=True)):
        img, rad, g = _run(sc, pipe)
        assert np.abs(img - img0).max() < 2e-5, (pipe, float(np.abs(img - img0).max()))
        if not pipe.compute_cov3D_python:
            np.testing.assert_array_equal(rad, rad0)
        else:       # torch builds Sigma in another operation order: a radius may round across an integer on a handful of splats
            assert (rad != rad0).mean() < 2e-3
        for k in g0:
            assert k in g, (pipe, k)
            assert grad_err(g[k], g0[k]) < GRAD_RTOL, (pipe, k, grad_err(g[k], g0[k]))


def test_override_color_branch_against_the_oracle():
    sc = synth.make_scene(P=2000, width=96, height=64, sh_degree=2, s0=0.05, seed=32, bg=(0.0, 0.3, 0.1))
    col = np.random.default_rng(5).uniform(0.0, 1.0, (sc.P, 3)).astype(np.float32)
    img, rad, g = _run(sc, PipelineParams(), override=col)
    S = oracle_scene(sc, shs=None, colors_precomp=col)
    r = ref.get("f32")
    f = r.forward(S)
    dL = sc.dL_dimage * (3.0 * S.H * S.W)
    gb = r.backward(f, dL.astype(np.float32))
    np.testing.assert_array_equal(rad, f["radii"])
    assert_image_close(img, f["color"])
    assert grad_err(g["override"], gb["dL_dcolors"]) < GRAD_RTOL
    assert grad_err(g["xyz"], gb["dL_dmeans3D"]) < GRAD_RTOL
    assert grad_err(g["means2D"], gb["dL_dmeans2D"]) < GRAD_RTOL
    assert "f_dc" not in g and "f_rest" not in g          # the SH features take no part
