"""-m gpu: the segmented, persistent reverse compositing pass (csrc/composite_bwd.hip: checkpoints written by the forward pass every
`segment_entries` list entries, work units drawn longest first) against the classic one-wave-per-half-tile kernel and against the
oracle, on scenes whose lists are long and whose pixels do not saturate -- the case it exists for."""
import ctypes as C

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import _lib, synth
from oracle import ref
from tests.helpers import GRAD_RTOL, assert_image_close, grad_err, hip_forward_backward, oracle_scene

pytestmark = pytest.mark.gpu

GRADS = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
         ("opacities", "dL_dopacity")]


def long_list_scene(P=20000, width=64, height=64, seed=5, opacity_scale=0.03, s0=0.02):
    """Thousands of faint splats per tile: nothing saturates, every pixel walks its whole list."""
    sc = synth.make_scene(P=P, width=width, height=height, sh_degree=1, s0=s0, seed=seed, bg=(0.3, 0.2, 0.1))
    sc.opacities = (sc.opacities * opacity_scale + 0.004).astype(np.float32)
    return sc


@pytest.fixture
def options():
    saved = {k: _lib.get_option(k) for k in ("persistent_bwd", "segment_entries")}
    yield
    for k, v in saved.items():
        _lib.set_option(k, v)


def _segments_summary(S):
    """Renders S forward + backward through the backend object and returns gsr_debug_read_segments' words."""
    from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, get_backend
    be = get_backend()
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda") if a is not None else torch.empty(0, device="cuda")
    P = int(np.asarray(S.means3D).shape[0])
    rs = GaussianRasterizationSettings(S.H, S.W, S.tanfovx, S.tanfovy, t(S.bg), S.scale_modifier, t(np.asarray(S.viewmatrix).reshape(4, 4)),
                                       t(np.asarray(S.projmatrix).reshape(4, 4)), S.sh_degree, t(S.campos), False, False)
    args = (t(S.means3D), t(S.shs), t(S.colors_precomp), t(np.asarray(S.opacities).reshape(P, 1)), t(S.scales), t(S.rotations), t(S.cov3D_precomp))
    n, color, radii, geom, binning, img = be.forward(rs, *args)
    be.backward(rs, n, torch.ones_like(color), args[0], radii, args[1], args[2], args[4], args[5], args[6], geom, binning, img)
    out = np.zeros(8, np.uint32)
    _lib.check(be.lib.gsr_debug_read_segments(torch.cuda.current_stream().cuda_stream, S.W, S.H, img.data_ptr(), out.ctypes.data), "read segments")
    return dict(seg=int(out[0]), units=int(out[1]), slots=int(out[2]), pool=int(out[3]), half_tiles=int(out[4]), drawn=int(out[5]), n=n)


@pytest.mark.parametrize("seg", [64, 256])
def test_segmented_reverse_pass_matches_classic_kernel_and_oracle(options, seg):
    sc = long_list_scene()
    S = oracle_scene(sc)
    dL = np.random.default_rng(3).normal(size=(3, S.H, S.W)).astype(np.float32)
    r = ref.get("f32")
    f = r.forward(S); g = r.backward(f, dL)
    _lib.set_option("persistent_bwd", 0); _lib.set_option("segment_entries", 0)
    classic = hip_forward_backward(S, dL)
    _lib.set_option("persistent_bwd", 1); _lib.set_option("segment_entries", seg)
    st = _segments_summary(S)
    assert st["seg"] == seg and st["slots"] > 0, st                       # checkpoints were taken ...
    assert st["units"] > st["half_tiles"], st                               # ... and the lists hold more pieces than there are half tiles
    assert st["drawn"] >= st["units"], st                                   # every unit was drawn
    h = hip_forward_backward(S, dL)
    np.testing.assert_array_equal(h["radii"], classic["radii"])
    # the image differs only by the order its colour sums are added in (per-segment sums)
    assert np.abs(h["color"] - classic["color"]).max() < 2e-6
    assert_image_close(h["color"], f["color"])
    for a, b in GRADS:
        ref_g = np.asarray(g[b]).reshape(h["grads"][a].shape)
        assert grad_err(h["grads"][a], classic["grads"][a]) < 2e-4, (a, grad_err(h["grads"][a], classic["grads"][a]))    # atomics' ordering noise
        assert grad_err(h["grads"][a], ref_g) < GRAD_RTOL, (a, grad_err(h["grads"][a], ref_g))


def test_checkpoint_pool_exhaustion_and_overlong_remainders(options):
    """Lists far longer than 8 segments on every tile: waves run out of checkpoints (7 per wave) and the pool runs out of slots
    (2 per half tile); the rest of such a list stays one long unit.  Same gradients."""
    sc = long_list_scene(P=30000, width=48, height=48, seed=9, s0=0.03)
    S = oracle_scene(sc)
    dL = np.random.default_rng(4).normal(size=(3, S.H, S.W)).astype(np.float32)
    _lib.set_option("persistent_bwd", 0); _lib.set_option("segment_entries", 0)
    classic = hip_forward_backward(S, dL)
    _lib.set_option("persistent_bwd", 1); _lib.set_option("segment_entries", 64)
    st = _segments_summary(S)
    assert st["n"] / (st["half_tiles"] / 2) > 64 * 12, st                  # lists of well over 8 segments
    assert st["slots"] == min(st["pool"], 2 * st["half_tiles"]), st       # every band used up its share of the pool (2 slots per half tile)
    h = hip_forward_backward(S, dL)
    assert np.abs(h["color"] - classic["color"]).max() < 2e-6
    for a, _ in GRADS:
        assert grad_err(h["grads"][a], classic["grads"][a]) < 2e-4, a


def test_backward_twice_draws_the_units_twice(options):
    """retain_graph: the second gsr_backward of one forward pass resets the ticket counters and rebuilds the lists."""
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer
    _lib.set_option("persistent_bwd", 1); _lib.set_option("segment_entries", 64)
    sc = long_list_scene(P=8000, width=48, height=32, seed=2)
    S = oracle_scene(sc)
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda").requires_grad_(g)
    P = sc.P
    means3D, opac, shs, scales, rots = t(S.means3D, True), t(np.asarray(S.opacities).reshape(P, 1), True), t(S.shs, True), t(S.scales, True), t(S.rotations, True)
    rs = GaussianRasterizationSettings(S.H, S.W, S.tanfovx, S.tanfovy, t(S.bg), 1.0, t(np.asarray(S.viewmatrix).reshape(4, 4)),
                                       t(np.asarray(S.projmatrix).reshape(4, 4)), S.sh_degree, t(S.campos), False, False)
    means2D = torch.zeros((P, 3), device="cuda", requires_grad=True)
    color, _ = GaussianRasterizer(rs)(means3D=means3D, means2D=means2D, shs=shs, opacities=opac, scales=scales, rotations=rots)
    dL = torch.tensor(np.random.default_rng(1).normal(size=(3, S.H, S.W)).astype(np.float32), device="cuda")
    g1 = torch.autograd.grad(color, [means3D, opac, shs], grad_outputs=dL, retain_graph=True)
    g2 = torch.autograd.grad(color, [means3D, opac, shs], grad_outputs=dL)
    for a, b in zip(g1, g2):
        assert grad_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-4


def test_options_changed_between_forward_and_backward_are_safe(options):
    """The forward pass ran without checkpoints; the persistent reverse kernel then takes whole half tiles."""
    sc = long_list_scene(P=8000, width=48, height=32, seed=6)
    S = oracle_scene(sc)
    dL = np.random.default_rng(8).normal(size=(3, S.H, S.W)).astype(np.float32)
    _lib.set_option("persistent_bwd", 0); _lib.set_option("segment_entries", 0)
    classic = hip_forward_backward(S, dL)
    _lib.set_option("persistent_bwd", 1)           # segment_entries stays 0: no checkpoints, no lengths
    h = hip_forward_backward(S, dL)
    for a, _ in GRADS:
        assert grad_err(h["grads"][a], classic["grads"][a]) < 2e-4, a
