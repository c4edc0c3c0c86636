"""-m gpu: the HIP path (through the Python API -> ctypes -> C ABI of include/gsr.h) against the
CPU restatement (oracle/) on the same seeded inputs.  Bars: discrete per-Gaussian outputs and the
sorted splat list bit-exact; RGB <= 1e-4 abs; gradients <= 1e-3 of the tensor's largest magnitude."""
import ctypes as C

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import synth
from oracle import ref
from tests.helpers import GRAD_RTOL, assert_image_close, grad_err, hip_forward_backward, oracle_scene

pytestmark = pytest.mark.gpu


def _stage_dump(S):
    """Forward through the backend object to keep the workspaces, then read them back."""
    from gaussian_transformer_amd import _lib
    from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, get_backend
    be = get_backend()
    dev = "cuda"
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev) if a is not None else torch.empty(0, device=dev)
    P = int(np.asarray(S.means3D).shape[0])
    rs = GaussianRasterizationSettings(S.H, S.W, S.tanfovx, S.tanfovy, t(S.bg), S.scale_modifier,
                                       t(np.asarray(S.viewmatrix).reshape(4, 4)), t(np.asarray(S.projmatrix).reshape(4, 4)),
                                       S.sh_degree, t(S.campos), False, False)
    n, color, radii, geom, binning, img = be.forward(rs, t(S.means3D), t(S.shs), t(S.colors_precomp),
                                                     t(np.asarray(S.opacities).reshape(P, 1)), t(S.scales), t(S.rotations),
                                                     t(S.cov3D_precomp))
    lib = be.lib
    stream = torch.cuda.current_stream().cuda_stream
    d = dict(depth=np.zeros(P, np.float32), xy=np.zeros((P, 2), np.float32), conic_o=np.zeros((P, 4), np.float32),
             rgb=np.zeros((P, 3), np.float32), tiles=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8))
    _lib.check(lib.gsr_debug_read_geom(stream, P, geom.data_ptr(), *[d[k].ctypes.data for k in
                                                                      ("depth", "xy", "conic_o", "rgb", "tiles", "clamped")]), "read geom")
    T = ((S.W + 15) // 16) * ((S.H + 15) // 16)
    keys = np.zeros(max(n, 1), np.uint64); pl = np.zeros(max(n, 1), np.uint32); ranges = np.zeros((T, 2), np.uint32)
    _lib.check(lib.gsr_debug_read_binning(stream, n, S.W, S.H, binning.data_ptr() if n else None, img.data_ptr(),
                                          keys.ctypes.data, pl.ctypes.data, ranges.ctypes.data), "read binning")
    fT = np.zeros((S.H, S.W), np.float32); nc = np.zeros((S.H, S.W), np.uint32)
    _lib.check(lib.gsr_debug_read_image_state(stream, S.W, S.H, img.data_ptr(), fT.ctypes.data, nc.ctypes.data), "read image")
    return dict(n=n, color=color.cpu().numpy(), radii=radii.cpu().numpy(), keys=keys[:n], point_list=pl[:n], ranges=ranges,
                final_T=fT, n_contrib=nc, **d)


@pytest.fixture
def upstream_tile_rule():
    """Switch the exact tile culling off: the pair lists must then equal upstream's rule bit for bit."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("exact_tile_cull", 0)
    yield
    _lib.set_option("exact_tile_cull", 1)


@pytest.fixture(params=[2, 1], ids=["supertile_sort", "round1_lists"])
def lists_mode(request):
    """The two sort-free list builders: 2 = supertile_sort.hip (default), 1 = depth_order.hip + tile_lists.hip (round 1's path, which
    is also what the default falls back to when a super-tile's bin exceeds the LDS capacity)."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("tile_lists", request.param)
    yield request.param
    _lib.set_option("tile_lists", 2)


@pytest.fixture
def round1_lists():
    from gaussian_transformer_amd import _lib
    _lib.set_option("tile_lists", 1)
    yield
    _lib.set_option("tile_lists", 2)


# 0: one global sort, 1: rocPRIM depth sort + rocPRIM by-tile sort, 2: depth_order.hip + tile_lists.hip, 3: supertile_sort.hip
@pytest.mark.parametrize("two_level", [0, 1, 2, 3])
@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=1500, width=100, height=57, sh_degree=1, s0=0.06, seed=1, zmin=0.05, zmax=5.0),   # culled splats, ragged tiles
    dict(P=500, width=33, height=47, sh_degree=0, s0=0.5, seed=2),                            # huge splats: every tile
    dict(P=3000, width=40, height=24, sh_degree=0, s0=0.6, seed=4),      # ~3000 pairs per tile: 4096-element LDS class
    dict(P=20000, width=32, height=32, sh_degree=0, s0=0.8, seed=5),     # ~20000 pairs per tile: beyond LDS, rank-sort path
])
def test_stages_bit_exact_against_oracle(kw, two_level, upstream_tile_rule):
    from gaussian_transformer_amd import _lib
    _lib.set_option("two_level_sort", 1 if two_level else 0)
    _lib.set_option("tile_lists", {0: 0, 1: 0, 2: 1, 3: 2}[two_level])
    _lib.set_option("depth_buckets", 1 if two_level >= 2 else 0)
    try:
        _check_stages_bit_exact(kw, two_level)
    finally:
        _lib.set_option("two_level_sort", 1)
        _lib.set_option("tile_lists", 2)
        _lib.set_option("depth_buckets", 1)


@pytest.mark.parametrize("kw", [
    dict(P=20000, width=3840, height=40, sh_degree=0, s0=0.02, seed=11),         # 30 x 1 super-tiles, ragged right edge
    dict(P=4000, width=200, height=1100, sh_degree=0, s0=0.05, seed=12),         # 2 x 9 super-tiles, tall splats
    dict(P=300, width=700, height=500, sh_degree=0, s0=1.5, seed=13),            # splats over dozens of super-tiles
    dict(P=70000, width=640, height=360, sh_degree=0, s0=0.01, seed=14),         # 137 level-1 workgroups, many segments
    dict(P=3000, width=8300, height=130, sh_degree=0, s0=0.02, seed=15),         # 65 x 2 super-tiles... still <= 512
    dict(P=3000, width=8300, height=1100, sh_degree=0, s0=0.02, seed=16),        # 585 super-tiles > 512: rocPRIM path
    dict(P=6000, width=64, height=64, sh_degree=0, s0=0.5, seed=41, zmin=4.0, zmax=4.0),     # one bin of 6000 entries, ONE depth
    dict(P=6500, width=64, height=64, sh_degree=0, s0=0.3, seed=42, zmin=0.21, zmax=40.0),   # one bin, depths over 7 octaves
    dict(P=13000, width=64, height=64, sh_degree=0, s0=0.3, seed=43, zmin=0.21, zmax=40.0),  # 7168 < bin <= 14336: one workgroup per CU
])
def test_tile_lists_bit_exact(kw, upstream_tile_rule, lists_mode):
    """Both sort-free list builders against the oracle's sorted pair list, per tile, on grids that stress the super-tile
    geometry; the last case exceeds round 1's LDS lane-mask capacity (that path must take the sort path by itself)."""
    _check_stages_bit_exact(dict(kw), 2)


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=1500, width=100, height=57, sh_degree=1, s0=0.06, seed=1, zmin=0.05, zmax=5.0),   # culled splats
    dict(P=20000, width=32, height=32, sh_degree=0, s0=0.8, seed=5),
    dict(P=100000, width=320, height=200, sh_degree=0, s0=0.01, seed=6),                     # 25 counting workgroups
    dict(P=6000, width=160, height=112, sh_degree=0, s0=0.03, seed=7, zmin=4.0, zmax=4.0),   # one depth: bitonic path
    dict(P=20000, width=160, height=112, sh_degree=0, s0=0.02, seed=8, zmin=4.0, zmax=4.0),  # bucket > LDS: rocPRIM fallback
    dict(P=30000, width=160, height=112, sh_degree=0, s0=0.02, seed=9, quantize_z=0.5),      # 17 depths: ties in every bucket
    dict(P=30000, width=160, height=112, sh_degree=0, s0=0.02, seed=10, quantize_z=0.001),   # small tie groups: rank path
])
def test_bucketed_depth_order_bit_exact(kw, upstream_tile_rule, round1_lists):
    """depth_order.hip forced on for every P (automatic from P = 1024): same sorted pair list as the oracle's one
    stable sort, including exact depth ties (ascending Gaussian id) and degenerate depth distributions."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("depth_buckets", 2)
    try:
        _check_stages_bit_exact(dict(kw), 1)
    finally:
        _lib.set_option("depth_buckets", 1)
        _lib.set_option("depth_log_map", 0)        # an overflowing case switches the process to the log map: undo


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=0, s0=0.03, seed=0),
    dict(P=100000, width=320, height=200, sh_degree=0, s0=0.01, seed=6),
    dict(P=30000, width=160, height=112, sh_degree=0, s0=0.02, seed=9, quantize_z=0.5),
    dict(P=6000, width=160, height=112, sh_degree=0, s0=0.03, seed=7, zmin=4.0, zmax=4.0),
])
def test_depth_order_log_bucket_map_bit_exact(kw, upstream_tile_rule, round1_lists):
    """The bucket map that is linear in the depth BITS (what the library switches to after an overflow)."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("depth_log_map", 1)
    try:
        _check_stages_bit_exact(dict(kw), 2)
    finally:
        _lib.set_option("depth_log_map", 0)


def test_depth_outliers_switch_the_bucket_map(upstream_tile_rule, round1_lists):
    """A handful of far outliers stretch the linear depth map until the whole scene shares one bucket: that frame takes
    the rocPRIM path (same lists), the library switches to the log map and the next frame is bucketed again."""
    from gaussian_transformer_amd import _lib
    kw = dict(P=60000, width=320, height=200, sh_degree=0, s0=0.01, seed=21, zmin=3.0, zmax=6.0, outliers=20)
    assert _lib.get_option("depth_log_map") == 0
    try:
        _check_stages_bit_exact(dict(kw), 2)
        assert _lib.get_option("depth_log_map") == 1
        _check_stages_bit_exact(dict(kw), 2)
    finally:
        _lib.set_option("depth_log_map", 0)


@pytest.mark.parametrize("kw", [
    dict(P=1, width=64, height=48, sh_degree=0, s0=0.3, seed=31),                         # one splat
    dict(P=40, width=17, height=9, sh_degree=0, s0=0.2, seed=32),                         # one ragged tile, one super-tile
    dict(P=5000, width=3840, height=2160, sh_degree=0, s0=0.01, seed=33, giants=3),       # 510 super-tiles; splats over all of them
    dict(P=2000, width=1920, height=1080, sh_degree=0, s0=0.2, seed=34),                  # every rectangle beyond 8 x 15 tiles
])
def test_tile_list_edge_cases_bit_exact(kw, upstream_tile_rule, lists_mode):
    """Forced through depth_order.hip + tile_lists.hip whatever P is: degenerate sizes, and rectangles too large for the
    packed row spans (the level-1 placement then evaluates the ellipse itself, one lane walking hundreds of super-tiles)."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("depth_buckets", 2)
    try:
        _check_stages_bit_exact(dict(kw), 2)
    finally:
        _lib.set_option("depth_buckets", 1)


def test_tile_list_edge_cases_with_exact_culling(lists_mode):
    """Same degenerate scenes in the default mode (exact culling on): image against the oracle, N never above the upstream count."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("depth_buckets", 2)
    try:
        for kw in (dict(P=1, width=64, height=48, sh_degree=0, s0=0.3, seed=31), dict(P=40, width=17, height=9, sh_degree=0, s0=0.2, seed=32),
                   dict(P=5000, width=3840, height=2160, sh_degree=0, s0=0.01, seed=33, giants=3)):
            kw = dict(kw)
            ng = kw.pop("giants", 0)
            sc = synth.make_scene(**kw)
            if ng:
                sc.scales[:ng] = np.array([8.0, 5.0, 0.5], np.float32)
            S = oracle_scene(sc)
            f = ref.get("f32").forward(S)
            h = _stage_dump(S)
            np.testing.assert_array_equal(h["radii"], f["radii"])
            assert h["n"] <= f["num_rendered"]
            assert_image_close(h["color"], f["color"])
    finally:
        _lib.set_option("depth_buckets", 1)


def _check_stages_bit_exact(kw, two_level):
    qz = kw.pop("quantize_z", None) if "quantize_z" in kw else None
    nout = kw.pop("outliers", 0) if "outliers" in kw else 0
    ng = kw.pop("giants", 0) if "giants" in kw else 0
    sc = synth.make_scene(**kw)
    if ng:                                     # a few splats as large as the whole view
        sc.scales[:ng] = np.array([8.0, 5.0, 0.5], np.float32)
    if qz:
        sc.means3D[:, 2] = np.round(sc.means3D[:, 2] / qz) * qz
    if nout:                                   # same screen position, 100x - 1000x the depth
        f = np.logspace(2, 3, nout).astype(np.float32)
        sc.means3D[:nout] *= f[:, None]
        sc.scales[:nout] *= f[:, None]
    S = oracle_scene(sc, scale_modifier=1.0)
    f = ref.get("f32").forward(S)
    og, ob, oi = f["state"].geom(), f["state"].binning(), f["state"].image_state()
    h = _stage_dump(S)
    # S1-S6: discrete outputs exact, continuous outputs bit-equal (same op order, contraction off)
    np.testing.assert_array_equal(h["radii"], f["radii"])
    np.testing.assert_array_equal(h["tiles"], og["tiles_touched"])
    vis = f["radii"] > 0
    np.testing.assert_array_equal(h["depth"][vis], og["depth"][vis])
    np.testing.assert_array_equal(h["xy"][vis], og["xy"][vis])
    np.testing.assert_array_equal(h["conic_o"][vis], og["conic_o"][vis])
    np.testing.assert_allclose(h["rgb"][vis], og["rgb"][vis], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(h["clamped"][vis].astype(bool), og["clamped"][vis].astype(bool))
    # S7-S8: identical sorted list (stable sort => deterministic), identical ranges
    assert h["n"] == f["num_rendered"]
    if not two_level:                      # the 64-bit tile<<32|depth keys only exist in global-sort mode
        np.testing.assert_array_equal(h["keys"], ob["keys"])
    np.testing.assert_array_equal(h["point_list"], ob["vals"])
    np.testing.assert_array_equal(h["ranges"], ob["ranges"])
    # S9
    assert_image_close(h["color"], f["color"])
    assert np.abs(h["final_T"] - oi["final_T"]).max() < 1e-4 or (np.abs(h["final_T"] - oi["final_T"]) > 1e-4).mean() < 2e-4
    assert (h["n_contrib"] != oi["n_contrib"]).mean() < 1e-3


@pytest.mark.parametrize("kw", [
    dict(P=20000, width=1280, height=720, sh_degree=0, s0=0.05, seed=61),      # all three rectangle classes of supertile_sort.hip
    dict(P=9000, width=1280, height=720, sh_degree=0, s0=0.12, seed=62),       # mostly medium rectangles: the LDS record list overflows
    dict(P=3000, width=1920, height=1080, sh_degree=0, s0=0.3, seed=63),       # mostly large ones (one wave each)
    dict(P=30000, width=333, height=517, sh_degree=0, s0=0.02, seed=64),       # ragged right / bottom super-tiles
])
def test_exact_culling_lists_identical_across_builders(kw):
    """Default mode (exact culling on): the three list builders -- one global sort of the keys binning.hip emits, round 1's
    depth order + tile lists, and supertile_sort.hip fed by the masks preprocess prepares -- must produce the same
    point_list and ranges, entry for entry (the oracle has no list for this rule; the builders check each other)."""
    from gaussian_transformer_amd import _lib
    S = oracle_scene(synth.make_scene(**kw))
    dumps = {}
    try:
        for name, opts in (("global_sort", dict(two_level_sort=0, tile_lists=0, depth_buckets=0)),
                           ("round1_lists", dict(two_level_sort=1, tile_lists=1, depth_buckets=2)),
                           ("supertile_sort", dict(two_level_sort=1, tile_lists=2, depth_buckets=1))):
            for k, v in opts.items():
                _lib.set_option(k, v)
            dumps[name] = _stage_dump(S)
    finally:
        _lib.set_option("two_level_sort", 1); _lib.set_option("tile_lists", 2); _lib.set_option("depth_buckets", 1)
    a = dumps["global_sort"]
    assert a["n"] > 0
    for name in ("round1_lists", "supertile_sort"):
        b = dumps[name]
        assert b["n"] == a["n"], name
        np.testing.assert_array_equal(b["ranges"], a["ranges"], err_msg=name)
        np.testing.assert_array_equal(b["point_list"], a["point_list"], err_msg=name)
        np.testing.assert_array_equal(b["color"], a["color"], err_msg=name)


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=1500, width=100, height=57, sh_degree=1, s0=0.06, seed=1, zmin=0.05, zmax=5.0),
    dict(P=500, width=33, height=47, sh_degree=0, s0=0.5, seed=2),
    dict(P=2000, width=256, height=144, sh_degree=0, s0=0.02, seed=3),
])
def test_exact_tile_culling_keeps_every_pair_a_pixel_can_blend(kw):
    """Default mode: the emitted list must be a subsequence of upstream's list (same order) that still
    contains every (tile, Gaussian) pair for which some pixel of the tile passes upstream's tests
    (power <= 0 and alpha >= 1/255); then no output can change."""
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    f = ref.get("f32").forward(S)
    og, ob = f["state"].geom(), f["state"].binning()
    h = _stage_dump(S)
    np.testing.assert_array_equal(h["radii"], f["radii"])            # radii keep upstream's meaning
    assert h["n"] <= f["num_rendered"]
    gridx = (S.W + 15) // 16
    xy, co = og["xy"].astype(np.float64), og["conic_o"].astype(np.float64)
    kept = 0
    for t in range(ob["ranges"].shape[0]):
        full = ob["vals"][ob["ranges"][t, 0]:ob["ranges"][t, 1]]
        mine = h["point_list"][h["ranges"][t, 0]:h["ranges"][t, 1]]
        pos = {int(g): i for i, g in enumerate(full)}
        idx = [pos[int(g)] for g in mine]                              # KeyError = pair not in upstream's list
        assert idx == sorted(idx) and len(set(idx)) == len(idx)        # subsequence, same depth order
        if len(full) == 0:
            continue
        tx, ty = t % gridx, t // gridx
        xs = np.arange(tx * 16, min(tx * 16 + 16, S.W)); ys = np.arange(ty * 16, min(ty * 16 + 16, S.H))
        X, Y = np.meshgrid(xs, ys)
        dx = xy[full, 0][:, None, None] - X[None]; dy = xy[full, 1][:, None, None] - Y[None]
        power = -0.5 * (co[full, 0][:, None, None] * dx * dx + co[full, 2][:, None, None] * dy * dy) - co[full, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, co[full, 3][:, None, None] * np.exp(power))
        needed = ((power <= 0) & (alpha >= 1.0 / 255.0)).any(axis=(1, 2))
        assert set(full[needed].tolist()) <= set(mine.tolist()), f"tile {t}: a blendable pair was culled"
        kept += len(mine)
    assert kept == h["n"]
    assert_image_close(h["color"], f["color"])


CASES = [
    dict(scene=dict(P=2000, width=128, height=96, sh_degree=3, s0=0.03, seed=10), over=dict()),
    dict(scene=dict(P=1000, width=100, height=57, sh_degree=2, s0=0.05, seed=11, bg=(0.4, 0.1, 0.9)), over=dict(scale_modifier=0.7)),
    dict(scene=dict(P=800, width=64, height=64, sh_degree=1, s0=0.1, seed=12, zmin=0.05, zmax=4.0, bg=(1, 1, 1)), over=dict()),
    dict(scene=dict(P=600, width=50, height=40, sh_degree=0, s0=0.2, seed=13, tanfovx=0.05), over=dict()),   # FoV clamp active
    dict(scene=dict(P=3000, width=256, height=256, sh_degree=3, s0=0.01, seed=14, max_sh_degree=3), over=dict()),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_forward_backward_parity(case):
    c = CASES[case]
    sc = synth.make_scene(**c["scene"])
    S = oracle_scene(sc, **c["over"])
    rng = np.random.default_rng(100 + case)
    dL = rng.normal(size=(3, S.H, S.W)).astype(np.float32)
    r = ref.get("f32")
    f = r.forward(S); g = r.backward(f, dL)
    h = hip_forward_backward(S, dL)
    np.testing.assert_array_equal(h["radii"], f["radii"])
    assert_image_close(h["color"], f["color"])
    hg = h["grads"]
    pairs = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
             ("rotations", "dL_drots")]
    for a, b in pairs:
        assert grad_err(hg[a], g[b]) < GRAD_RTOL, (a, grad_err(hg[a], g[b]))
    assert grad_err(hg["opacities"].reshape(-1), g["dL_dopacity"]) < GRAD_RTOL
    assert np.all(hg["means2D"][:, 2] == 0)


def test_composited_mask_covers_every_gaussian_with_a_gradient():
    """gsr_composited_mask: a superset of the Gaussians that receive a gradient, a subset of radii > 0 -- and in a dense, mostly
    occluded cloud much smaller than that (what a data-parallel trainer exchanges)."""
    from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, composited_mask
    sc = synth.make_scene(P=60000, width=320, height=200, sh_degree=1, s0=0.03, seed=81)       # dense: most of it is occluded
    S = oracle_scene(sc)
    dev = "cuda"
    t = lambda a, g=False: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev).requires_grad_(g)
    P = sc.P
    ps = [t(S.means3D, True), t(np.asarray(S.opacities).reshape(P, 1), True), t(S.shs, True), t(S.scales, True), t(S.rotations, True)]
    rs = GaussianRasterizationSettings(S.H, S.W, S.tanfovx, S.tanfovy, t(S.bg), S.scale_modifier, t(np.asarray(S.viewmatrix).reshape(4, 4)),
                                       t(np.asarray(S.projmatrix).reshape(4, 4)), S.sh_degree, t(S.campos), False, False)
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    color, radii = GaussianRasterizer(raster_settings=rs)(means3D=ps[0], means2D=m2, shs=ps[2], opacities=ps[1], scales=ps[3], rotations=ps[4])
    mask = composited_mask()
    assert mask is not None and mask.shape == (P,) and mask.dtype == torch.bool
    dL = torch.tensor(np.random.default_rng(3).normal(size=(3, S.H, S.W)).astype(np.float32), device=dev)
    g = torch.autograd.grad(color, ps + [m2], grad_outputs=dL)
    nz = torch.zeros(P, dtype=torch.bool, device=dev)
    for x in g:
        nz |= x.reshape(P, -1).abs().sum(1) > 0
    assert not bool((nz & ~mask).any()), "a Gaussian outside the mask received a gradient"
    assert not bool((mask & ~(radii > 0)).any())
    assert int(mask.sum()) < int((radii > 0).sum())                  # occlusion does make it smaller here
    assert int(nz.sum()) > 0


def test_backward_with_replica_accumulator_rows():
    """Splats over hundreds of tiles: their waves add into replica rows of the gradient accumulator (supertile_sort.hip hands
    them out, composite_bwd.hip picks replica tile mod K, pergauss_bwd.hip folds them).  Same gradients as the oracle."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(P=4000, width=960, height=540, sh_degree=1, s0=0.02, seed=71)
    sc.scales[:6] = np.array([[6.0, 4.0, 0.4], [3.0, 3.0, 0.3], [2.0, 0.2, 2.0], [0.8, 1.5, 0.5], [5.0, 0.3, 0.3], [1.2, 1.2, 1.2]], np.float32)
    sc.opacities[:6] = 0.6
    S = oracle_scene(sc)
    dL = np.random.default_rng(171).normal(size=(3, S.H, S.W)).astype(np.float32)
    r = ref.get("f32")
    f = r.forward(S); g = r.backward(f, dL)
    assert (f["state"].geom()["tiles_touched"][:6] >= 256).sum() >= 3         # the scene does contain such splats
    h = hip_forward_backward(S, dL)
    np.testing.assert_array_equal(h["radii"], f["radii"])
    assert_image_close(h["color"], f["color"])
    hg = h["grads"]
    for a, b in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"), ("rotations", "dL_drots")]:
        assert grad_err(hg[a], g[b]) < GRAD_RTOL, (a, grad_err(hg[a], g[b]))
        # the giants themselves (their rows are the ones that went through replicas)
        big = np.abs(g[b][:6]).max() + 1e-30
        assert np.abs(hg[a][:6] - g[b][:6].reshape(hg[a][:6].shape)).max() / big < GRAD_RTOL, a
    assert grad_err(hg["opacities"].reshape(-1), g["dL_dopacity"]) < GRAD_RTOL
    # deterministic mode (no atomics, no replicas in use) agrees too
    _lib.set_option("deterministic_bwd", 1)
    try:
        hd = hip_forward_backward(S, dL)["grads"]
    finally:
        _lib.set_option("deterministic_bwd", 0)
    for a in ("means3D", "scales", "rotations", "shs"):
        assert grad_err(hd[a], hg[a]) < 5e-4, a


def _table_scene(width=1008, height=567):
    """The reference's table_ds SfM cloud (tests/golden/table_points3D.ply, 17 618 points) turned into Gaussians the way
    scene/gaussian_model.py:124-146 initialises them (scale = sqrt of the mean squared distance to the 3 nearest points,
    identity rotation, opacity 0.1, SH dc from the colour), seen by table_ds's own camera intrinsics (cameras.bin,
    f = 3049.78 on 4032 x 2268, at 1/4 resolution) from the COLMAP origin.  images.bin (the poses) is not in the snapshot."""
    import os
    from scipy.spatial import cKDTree
    from gaussian_transformer_amd import io as gio
    pc = gio.fetch_point_cloud(os.path.join(os.path.dirname(__file__), "golden", "table_points3D.ply"))
    xyz = pc.points.astype(np.float32)
    d, _ = cKDTree(xyz).query(xyz, k=4)
    dist2 = np.maximum((d[:, 1:] ** 2).mean(1), 1e-7)
    P = xyz.shape[0]
    scales = np.repeat(np.sqrt(dist2)[:, None], 3, axis=1).astype(np.float32)
    rots = np.zeros((P, 4), np.float32); rots[:, 0] = 1
    opac = np.full((P, 1), 0.1, np.float32)
    shs = np.zeros((P, 16, 3), np.float32)
    shs[:, 0, :] = (pc.colors - 0.5) / 0.28209479177387814
    cam = synth.identity_camera(width, height, tanfovx=4032.0 / (2.0 * 3049.779011853469))
    dL = (np.random.default_rng(5).normal(size=(3, height, width)) / (3.0 * height * width)).astype(np.float32)
    return synth.SyntheticScene(cam, xyz, scales, rots, opac, shs, 3, np.zeros(3, np.float32), dL)


def test_table_scene_lists_bit_exact(upstream_tile_rule, lists_mode):
    """Real SfM depth / footprint distribution through depth_order.hip + tile_lists.hip: same per-tile lists as the oracle."""
    sc = _table_scene()
    S = oracle_scene(sc)
    f = ref.get("f32").forward(S)
    ob = f["state"].binning()
    h = _stage_dump(S)
    np.testing.assert_array_equal(h["radii"], f["radii"])
    assert (f["radii"] > 0).sum() > 10000 and f["num_rendered"] > 100000      # the cloud is in view (10 726 splats, 116 814 pairs)
    assert h["n"] == f["num_rendered"]
    np.testing.assert_array_equal(h["point_list"], ob["vals"])
    np.testing.assert_array_equal(h["ranges"], ob["ranges"])
    assert_image_close(h["color"], f["color"])


def test_table_scene_forward_backward_parity():
    sc = _table_scene()
    S = oracle_scene(sc)
    dL = np.random.default_rng(6).normal(size=(3, S.H, S.W)).astype(np.float32)
    r = ref.get("f32")
    f = r.forward(S); g = r.backward(f, dL)
    h = hip_forward_backward(S, dL)
    np.testing.assert_array_equal(h["radii"], f["radii"])
    assert_image_close(h["color"], f["color"])
    hg = h["grads"]
    for a, b in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
                 ("rotations", "dL_drots")]:
        assert grad_err(hg[a], g[b]) < GRAD_RTOL, (a, grad_err(hg[a], g[b]))
    assert grad_err(hg["opacities"].reshape(-1), g["dL_dopacity"]) < GRAD_RTOL


@pytest.mark.parametrize("npx", [1, 2, 4])
def test_backward_variants_blocks_per_wave(npx):
    """The reverse compositing kernel is instantiated for 1, 2 and 4 8x8 blocks per wave."""
    from gaussian_transformer_amd import _lib
    _lib.set_option("bwd_blocks_per_wave", npx)
    _lib.set_option("fwd_blocks_per_wave", npx)
    try:
        for case in (1, 4):
            c = CASES[case]
            sc = synth.make_scene(**c["scene"])
            S = oracle_scene(sc, **c["over"])
            dL = np.random.default_rng(7).normal(size=(3, S.H, S.W)).astype(np.float32)
            r = ref.get("f32")
            f = r.forward(S); g = r.backward(f, dL)
            h = hip_forward_backward(S, dL)
            assert_image_close(h["color"], f["color"])
            for a, b in [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
                         ("rotations", "dL_drots")]:
                assert grad_err(h["grads"][a], g[b]) < GRAD_RTOL, (npx, case, a)
            assert grad_err(h["grads"]["opacities"].reshape(-1), g["dL_dopacity"]) < GRAD_RTOL
    finally:
        _lib.set_option("bwd_blocks_per_wave", 2)
        _lib.set_option("fwd_blocks_per_wave", 2)


def test_precomputed_colour_and_covariance_parity():
    sc = synth.make_scene(P=1200, width=96, height=80, sh_degree=0, s0=0.05, seed=21, bg=(0.2, 0.3, 0.4))
    rng = np.random.default_rng(2)
    A = rng.normal(size=(sc.P, 3, 3)) * 0.05
    cov = A @ np.transpose(A, (0, 2, 1)) + 1e-4 * np.eye(3)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
    colors = rng.uniform(0, 1, size=(sc.P, 3))
    S = oracle_scene(sc, shs=None, colors_precomp=colors, scales=None, rotations=None, cov3D_precomp=cov6)
    dL = rng.normal(size=(3, S.H, S.W)).astype(np.float32)
    r = ref.get("f32")
    f = r.forward(S); g = r.backward(f, dL)
    h = hip_forward_backward(S, dL)
    assert_image_close(h["color"], f["color"])
    assert grad_err(h["grads"]["colors_precomp"], g["dL_dcolors"]) < GRAD_RTOL
    assert grad_err(h["grads"]["cov3D_precomp"], g["dL_dcov3D"]) < GRAD_RTOL
    assert grad_err(h["grads"]["means3D"], g["dL_dmeans3D"]) < GRAD_RTOL
    assert h["grads"]["shs"] is None and h["grads"]["scales"] is None


def test_edge_cases_empty_culled_and_background():
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    cam = synth.identity_camera(40, 24)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    rs = GaussianRasterizationSettings(24, 40, cam.tanfovx, cam.tanfovy, t([0.2, 0.4, 0.6]), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), 0, t(cam.camera_center), False, False)
    rast = GaussianRasterizer(raster_settings=rs)
    # P = 0 -> zeros (not background), radii empty
    z = lambda *s: torch.zeros(*s, device=dev)
    color, radii = rast(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, 24, 40) and radii.shape == (0,) and float(color.abs().max()) == 0.0
    # everything culled -> background everywhere, gradients all zero
    m = t([[0, 0, 0.1], [0, 0, -2.0]]).requires_grad_(True)
    m2 = z(2, 3).requires_grad_(True)
    color, radii = rast(means3D=m, means2D=m2, opacities=t([[0.9], [0.9]]), colors_precomp=t(np.ones((2, 3))),
                        scales=t(np.full((2, 3), 0.1)), rotations=t([[1, 0, 0, 0]] * 2))
    assert int(radii.abs().sum()) == 0
    np.testing.assert_allclose(color[:, 3, 5].detach().cpu().numpy(), [0.2, 0.4, 0.6], atol=1e-7)
    color.sum().backward()
    assert float(m.grad.abs().max()) == 0.0 and float(m2.grad.abs().max()) == 0.0
    # no_grad path
    with torch.no_grad():
        c2, _ = rast(means3D=m, means2D=m2, opacities=t([[0.9], [0.9]]), colors_precomp=t(np.ones((2, 3))),
                     scales=t(np.full((2, 3), 0.1)), rotations=t([[1, 0, 0, 0]] * 2))
    assert not c2.requires_grad
    vis = rast.markVisible(m.detach())
    assert vis.tolist() == [False, False]


def test_strided_nonleaf_inputs_and_validation_errors():
    """train_transformer.py:40-50 feeds strided views of a 26-wide row."""
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = "cuda"
    sc = synth.make_scene(P=700, width=64, height=48, sh_degree=1, s0=0.08, seed=31, max_sh_degree=1)
    cam = sc.camera
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)
    P = sc.P
    row = torch.zeros((P, 26), device=dev)
    row[:, :12] = t(sc.shs.reshape(P, 12)); row[:, 12:16] = t(sc.rotations); row[:, 16:17] = t(sc.opacities)
    row[:, 17:20] = t(sc.means3D); row[:, 20:23] = t(sc.scales)
    row.requires_grad_(True)
    x = row * 1.0                                   # non-leaf
    rs = GaussianRasterizationSettings(48, 64, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), 1, t(cam.camera_center), False, False)
    rast = GaussianRasterizer(raster_settings=rs)
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True) + 0
    m2.retain_grad()
    color, radii = rast(means3D=x[:, 17:20], means2D=m2, opacities=x[:, 16:17], shs=x[:, :12].reshape(P, 4, 3),
                        scales=x[:, 20:23], rotations=x[:, 12:16])
    dL = torch.tensor(np.random.default_rng(0).normal(size=(3, 48, 64)).astype(np.float32), device=dev)
    (color * dL).sum().backward()
    S = oracle_scene(sc)
    r = ref.get("f32"); f = r.forward(S); g = r.backward(f, dL.cpu().numpy())
    assert_image_close(color.detach().cpu().numpy(), f["color"])
    gr = row.grad.cpu().numpy()
    assert grad_err(gr[:, 17:20], g["dL_dmeans3D"]) < GRAD_RTOL
    assert grad_err(gr[:, :12].reshape(P, 4, 3), g["dL_dsh"]) < GRAD_RTOL
    assert grad_err(gr[:, 12:16], g["dL_drots"]) < GRAD_RTOL
    assert grad_err(m2.grad.cpu().numpy(), g["dL_dmeans2D"]) < GRAD_RTOL
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(means3D=x[:, 17:20], means2D=m2, opacities=x[:, 16:17], scales=x[:, 20:23], rotations=x[:, 12:16])
    with pytest.raises(Exception, match="scale/rotation pair"):
        rast(means3D=x[:, 17:20], means2D=m2, opacities=x[:, 16:17], shs=x[:, :12].reshape(P, 4, 3), scales=x[:, 20:23])
    with pytest.raises(RuntimeError):               # CPU tensors: loud failure, no fallback
        rast(means3D=x[:, 17:20].cpu(), means2D=m2.cpu(), opacities=x[:, 16:17].cpu(), shs=x[:, :12].reshape(P, 4, 3).cpu(),
             scales=x[:, 20:23].cpu(), rotations=x[:, 12:16].cpu())


def test_gradient_arena_matches_separate_buffers():
    """gradient_arena: gsr_backward writes the parameter gradients into slices of one flat bucket."""
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer
    from gaussian_transformer_amd.rasterizer import arena_floats, gradient_arena
    sc = synth.make_scene(P=1500, width=96, height=64, sh_degree=2, s0=0.05, seed=41)
    cam = sc.camera
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda").requires_grad_(g)
    inp = dict(means3D=t(sc.means3D, True), shs=t(sc.shs, True), opacities=t(sc.opacities, True), scales=t(sc.scales, True),
               rotations=t(sc.rotations, True))
    rs = GaussianRasterizationSettings(64, 96, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), 2, t(cam.camera_center), False, False)
    dL = t(np.random.default_rng(0).normal(size=(3, 64, 96)))
    params = list(inp.values())
    c, _ = GaussianRasterizer(raster_settings=rs)(means2D=torch.zeros(1500, 3, device="cuda", requires_grad=True), **inp)
    g_ref = torch.autograd.grad(c, params, grad_outputs=dL)
    flat = torch.zeros(arena_floats(1500, 9), device="cuda")
    c, _ = GaussianRasterizer(raster_settings=rs)(means2D=torch.zeros(1500, 3, device="cuda", requires_grad=True), **inp)
    with gradient_arena(flat):
        g_ar = torch.autograd.grad(c, params, grad_outputs=dL)
    off = 0
    for a, b in zip(g_ar, g_ref):
        assert grad_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-4          # atomics: order-dependent last bits
        n = a.numel()
        assert a.data_ptr() == flat.data_ptr() + 4 * off                   # a view of the bucket, in parameter order
        off += n
    assert off == flat.numel()


def test_gradient_arena_around_forward_and_backward_is_zero_filled_by_the_forward_pass():
    """An arena entered around the forward call too: its slices are announced (gsr_backward_prefill) and -- with the dense per-Gaussian
    stage -- zero-filled beside the forward pass; the gradients must be the same views with the same values (deterministic reverse pass:
    bit for bit) as with the arena around backward only, whatever the arena held before; a different arena at backward time makes
    the backward call fill by itself."""
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
    from gaussian_transformer_amd.rasterizer import arena_floats, gradient_arena
    sc = synth.make_scene(P=120000, width=640, height=400, sh_degree=3, s0=0.01, seed=43)
    cam, P = sc.camera, sc.P
    t = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float32), device="cuda").requires_grad_(g)
    inp = dict(means3D=t(sc.means3D, True), shs=t(sc.shs, True), opacities=t(sc.opacities.reshape(P, 1), True), scales=t(sc.scales, True),
               rotations=t(sc.rotations, True))
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), 3, t(cam.camera_center), False, False)
    dL = t(sc.dL_dimage)
    params = list(inp.values())
    m2 = lambda: torch.zeros(P, 3, device="cuda", requires_grad=True)
    nan = lambda: torch.full((arena_floats(P, 16),), float("nan"), device="cuda")
    _lib.set_option("deterministic_bwd", 1); _lib.set_option("dense_pergauss", 1)
    try:
        a_ref = nan()
        c, _ = GaussianRasterizer(raster_settings=rs)(means2D=m2(), **inp)
        with gradient_arena(a_ref):
            g_ref = [g.clone() for g in torch.autograd.grad(c, params, grad_outputs=dL)]
        a_both = nan()
        with gradient_arena(a_both):
            c, _ = GaussianRasterizer(raster_settings=rs)(means2D=m2(), **inp)
            torch.cuda.synchronize()
            assert bool((a_both == 0).all())                       # zeroed by the forward call
            g_both = torch.autograd.grad(c, params, grad_outputs=dL)
        a_fwd, a_bwd = nan(), nan()
        with gradient_arena(a_fwd):
            c, _ = GaussianRasterizer(raster_settings=rs)(means2D=m2(), **inp)
        with gradient_arena(a_bwd):
            g_swapped = torch.autograd.grad(c, params, grad_outputs=dL)
    finally:
        _lib.set_option("deterministic_bwd", 0); _lib.set_option("dense_pergauss", 2)
    off = 0
    for r, b, w in zip(g_ref, g_both, g_swapped):
        assert b.data_ptr() == a_both.data_ptr() + 4 * off and w.data_ptr() == a_bwd.data_ptr() + 4 * off
        assert torch.equal(r, b) and torch.equal(r, w)
        off += r.numel()
    assert off == a_both.numel() and float(g_ref[0].abs().max()) > 0


@pytest.mark.parametrize("deg,max_deg", [(3, 3), (1, 3), (0, 1)])
def test_fused_raw_parameter_path_matches_activation_graph(deg, max_deg):
    """render_fused (raw parameters, activations and the SH cat inside the kernels) against
    render() whose inputs go through torch's exp / sigmoid / normalize / cat and their autograd."""
    from gaussian_transformer_amd.model import GaussianParams
    from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render, render_fused
    sc = synth.make_scene(P=3000, width=128, height=80, sh_degree=deg, s0=0.04, seed=51, max_sh_degree=max_deg, bg=(0.2, 0.1, 0.4))
    dev = "cuda"
    cam = TorchCamera(sc.camera, dev)
    bg = torch.tensor(sc.bg, device=dev)
    dL = torch.tensor(np.random.default_rng(2).normal(size=(3, 80, 128)).astype(np.float32), device=dev)
    out = []
    for fn in (render, render_fused):
        pc = GaussianParams.from_synthetic(sc, dev)
        pc._rotation.data *= 1.7                                   # un-normalised quaternions: the normalize Jacobian matters
        pkg = fn(cam, pc, PipelineParams(), bg, 0.9)
        (pkg["render"] * dL).sum().backward()
        out.append((pkg["render"].detach().cpu().numpy(), pkg["radii"].cpu().numpy(), pkg["viewspace_points"].grad.cpu().numpy(),
                    [p.grad.cpu().numpy() for p in pc.parameters()]))
    (img0, rad0, v0, g0), (img1, rad1, v1, g1) = out
    np.testing.assert_array_equal(rad0, rad1)
    assert_image_close(img1, img0, atol=2e-6, outlier_frac=1e-4, outlier_max=6e-3)
    assert grad_err(v1, v0) < 1e-4
    for a, b, name in zip(g1, g0, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        assert a.shape == b.shape
        assert grad_err(a, b) < 2e-4, name


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=100000, width=640, height=360, sh_degree=0, s0=0.02, seed=31),
    dict(P=300000, width=800, height=800, sh_degree=0, s0=0.01, seed=32),
])
def test_forward_and_reverse_pass_take_the_same_decisions(kw):
    """The reverse pass must skip exactly the pairs the forward pass skipped (upstream evaluates one expression in both
    kernels): both kernels call gsr_device.h's splat_power_log2 / splat_alpha, spelled with explicit fma / mul.  The
    instrumented kernels count the blending (pixel, splat) pairs of each pass -- round 1's two expressions differed in
    1-2 pairs out of 131 M on config 3."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    _lib.set_option("count_lanes", 1)
    try:
        _lib.read_lane_counters()
        hip_forward_backward(S, sc.dL_dimage)
        c = _lib.read_lane_counters()
    finally:
        _lib.set_option("count_lanes", 0)
    assert c["fwd"]["lanes_ok"] > 0
    assert c["fwd"]["lanes_ok"] == c["bwd"]["lanes_ok"], (c["fwd"]["lanes_ok"], c["bwd"]["lanes_ok"])


def test_composited_mask_belongs_to_its_render_and_undersized_fused_arena_is_refused():
    """composited_mask(image) reads the workspace of the render that produced `image` (two renders in flight: each its own mask),
    and the fused raw-parameter path checks the gradient arena against M = 1 + rest coefficients."""
    from gaussian_transformer_amd import _lib
    from gaussian_transformer_amd.model import GaussianParams
    from gaussian_transformer_amd.rasterizer import arena_floats, composited_mask, gradient_arena
    from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render, render_fused
    from gaussian_transformer_amd.camera import look_at_camera
    sc = synth.make_scene(P=20000, width=160, height=96, sh_degree=1, s0=0.03, seed=82, max_sh_degree=1)
    pc = GaussianParams.from_synthetic(sc, "cuda")
    bg = torch.tensor(sc.bg, device="cuda")
    cam_a = TorchCamera(sc.camera, "cuda")
    cam_b = TorchCamera(look_at_camera(np.array([1.5, 0.0, 0.0]), np.array([0.0, 0.0, 6.0]), (0.0, -1.0, 0.0), sc.camera.FoVx, 160, 96), "cuda")
    a = render(cam_a, pc, PipelineParams(), bg)
    mask_a_now = composited_mask()
    b = render(cam_b, pc, PipelineParams(), bg)
    mask_a, mask_b = composited_mask(a["render"]), composited_mask(b["render"])
    assert mask_a is not None and mask_b is not None
    assert bool((mask_a == mask_a_now).all())                   # a's mask is still a's after b was rendered
    assert bool((mask_a != mask_b).any())                        # another camera composites other Gaussians
    assert bool((composited_mask() == mask_b).all())             # without an argument: the most recent render
    # fused path: M + Mrest = 4 coefficients; an arena sized for the DC term only must be refused
    f = render_fused(cam_a, pc, PipelineParams(), bg)
    small = torch.zeros(arena_floats(sc.P, 1), device="cuda")
    with gradient_arena(small):
        with pytest.raises(_lib.GsrError, match="gradient arena"):
            f["render"].sum().backward()
    ok = torch.zeros(arena_floats(sc.P, 4), device="cuda")
    f = render_fused(cam_a, pc, PipelineParams(), bg)
    with gradient_arena(ok):
        f["render"].sum().backward()
    assert float(ok.abs().sum()) > 0


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=20000, width=250, height=131, sh_degree=1, s0=0.05, seed=7),          # ragged edge tiles, long lists, saturating pixels
    dict(P=100000, width=640, height=360, sh_degree=0, s0=0.02, seed=31),
    dict(P=300000, width=800, height=800, sh_degree=0, s0=0.01, seed=32),
])
def test_forward_walkers_agree(kw):
    """The default forward kernel walks a staged batch in hand-written gfx950 assembly (composite_fwd.hip::walk_batch_2blocks: the
    per-pair decisions narrow EXEC directly instead of going through scalar masks); the instrumented instantiations and
    `asm_walk = 0` use the C++ walk.  Same arithmetic, same operand order, same comparisons: colour, final transmittance, last
    contributor and the reachability bytes handed to the reverse pass must be identical, bit for bit; so must the gradients (the
    reverse pass sees the same forward state)."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    assert _lib.get_option("asm_walk") == 1
    a = _stage_dump(S)
    ga = hip_forward_backward(S, sc.dL_dimage)
    _lib.set_option("asm_walk", 0)
    try:
        b = _stage_dump(S)
        gb = hip_forward_backward(S, sc.dL_dimage)
    finally:
        _lib.set_option("asm_walk", 1)
    assert a["n"] == b["n"] and a["n"] > 0
    for k in ("color", "final_T", "n_contrib", "point_list", "ranges"):
        assert np.array_equal(a[k], b[k]), k
    assert (a["final_T"] < 1e-3).any() or kw["P"] < 5000          # the stop path ran
    assert np.array_equal(ga["color"], gb["color"])
    # the reverse pass adds with float atomics (order varies from run to run): same forward state -> same sums up to that order
    for k, v in ga["grads"].items():
        if v is not None:
            assert grad_err(v, gb["grads"][k]) <= 2e-4, k


@pytest.mark.parametrize("kw", [
    dict(P=3000, width=160, height=112, sh_degree=3, s0=0.03, seed=0),
    dict(P=20001, width=250, height=131, sh_degree=1, max_sh_degree=3, s0=0.05, seed=7),    # P not a multiple of 4: unaligned tensor tails in the zero-fill
    dict(P=100000, width=640, height=360, sh_degree=0, max_sh_degree=3, s0=0.02, seed=31),
    dict(P=300000, width=800, height=800, sh_degree=2, max_sh_degree=3, s0=0.003, seed=32),   # more Gaussians with a gradient than the record buffer holds
    dict(P=300000, width=800, height=800, sh_degree=2, s0=0.01, seed=32),                    # M = 9: not the dense variant's layout (streaming kernel both times)
])
def test_dense_per_gaussian_stage_matches_the_streaming_one(kw):
    """`dense_pergauss`: on the library's second stream, beside the compositing kernel, a fill kernel writes the zeros of every gradient
    output and a gather kernel lists the Gaussians with a gradient and copies their inputs into a compact buffer; pergauss_bwd then
    runs on full waves of those.  With the deterministic reverse pass (no atomics: identical accumulator rows) every gradient tensor
    must equal the streaming kernel's bit for bit -- and a buffer that held garbage before the call must hold exact zeros wherever
    no gradient exists."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    _lib.set_option("deterministic_bwd", 1)
    try:
        _lib.set_option("dense_pergauss", 0)
        a = hip_forward_backward(S, sc.dL_dimage)["grads"]
        _lib.set_option("dense_pergauss", 1)
        b = hip_forward_backward(S, sc.dL_dimage)["grads"]
    finally:
        _lib.set_option("deterministic_bwd", 0)
        _lib.set_option("dense_pergauss", 2)
    some = False
    for k, v in a.items():
        if v is None:
            assert b[k] is None
            continue
        assert np.array_equal(v, b[k]), (k, float(np.abs(v - b[k]).max()))
        some = some or np.abs(v).max() > 0
    assert some


@pytest.mark.parametrize("kw", [
    dict(P=20000, width=250, height=131, sh_degree=1, s0=0.05, seed=7),          # ragged edge tiles, long lists, saturating pixels
    dict(P=60000, width=320, height=200, sh_degree=0, s0=0.04, seed=11),         # several checkpoints per half tile
    dict(P=300000, width=800, height=800, sh_degree=0, s0=0.01, seed=32),
])
@pytest.mark.parametrize("long_n", [64, 256])
def test_forward_by_pairs_of_block_waves_agrees(kw, long_n):
    """On small images half tiles with long lists are walked by a workgroup of two waves, one per 8x8 block, which share the
    half tile's checkpoint slots and file its `info` together (composite_fwd.hip::fwd_unit_pair).  Colour, final transmittance,
    last contributor and lists must equal the one-wave-per-half-tile kernel's bit for bit; the gradients -- through the persistent,
    segmented reverse kernel that starts from those checkpoints -- within the order-of-addition tolerance."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    _lib.set_option("fwd_pair_long", 0)
    try:
        a = _stage_dump(S)
        ga = hip_forward_backward(S, sc.dL_dimage)
        _lib.set_option("fwd_pair_long", long_n)
        b = _stage_dump(S)
        gb = hip_forward_backward(S, sc.dL_dimage)
    finally:
        _lib.set_option("fwd_pair_long", -1)
    assert a["n"] == b["n"] and a["n"] > 0
    T = ((S.W + 15) // 16) * ((S.H + 15) // 16)
    assert (a["ranges"][:, 1] - a["ranges"][:, 0]).max() > long_n          # the pair path ran
    for k in ("color", "final_T", "n_contrib", "point_list", "ranges"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(ga["color"], gb["color"])
    for k, v in ga["grads"].items():
        if v is not None:
            assert grad_err(v, gb["grads"][k]) <= 2e-4, k


@pytest.mark.parametrize("kw", [
    dict(P=20000, width=250, height=131, sh_degree=1, s0=0.05, seed=7),
    dict(P=200000, width=1920, height=1080, sh_degree=3, s0=0.01, seed=3),       # > 6144 tiles: one workgroup per tile pair, not the persistent kernel
])
def test_compositing_with_several_waves_per_workgroup(kw):
    """`composite_waves_per_block` > 1 puts several waves' LDS slices into one workgroup: the assembly walks address theirs through a
    VGPR (forward) and through M0 + offset (the reverse pass's reduction rows) -- same image, same gradients as one wave per workgroup."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(**kw)
    S = oracle_scene(sc)
    a = hip_forward_backward(S, sc.dL_dimage)
    for wpb in (2, 4):
        _lib.set_option("composite_waves_per_block", wpb)
        try:
            b = hip_forward_backward(S, sc.dL_dimage)
        finally:
            _lib.set_option("composite_waves_per_block", 1)
        assert np.array_equal(a["color"], b["color"]), wpb
        for k, v in a["grads"].items():
            if v is not None:
                assert grad_err(v, b["grads"][k]) <= 2e-4, (wpb, k)


def test_reverse_pass_in_order_of_decreasing_length_on_large_images():
    """Above 6144 tiles the reverse compositing kernel takes its half tiles from per-band lists sorted by how far their pixels got
    (lengths filed by the forward pass, sorted by the planner that runs with the accumulator clearing): the same units, another
    order -- gradients equal up to the order of float additions; and a forward pass that filed nothing (option off at forward time)
    leaves the lists in tile order."""
    from gaussian_transformer_amd import _lib
    sc = synth.make_scene(P=200000, width=1920, height=1080, sh_degree=3, s0=0.01, seed=3)
    S = oracle_scene(sc)
    assert _lib.get_option("bwd_lpt") == 1
    a = hip_forward_backward(S, sc.dL_dimage)
    _lib.set_option("bwd_lpt", 0)
    try:
        b = hip_forward_backward(S, sc.dL_dimage)
    finally:
        _lib.set_option("bwd_lpt", 1)
    assert np.array_equal(a["color"], b["color"])
    for k, v in a["grads"].items():
        if v is not None:
            assert np.abs(v).max() > 0 or k == "means2D"
            assert grad_err(v, b["grads"][k]) <= 2e-4, k


@pytest.mark.parametrize("at", [1, 2, 0])
def test_announced_gradient_outputs_are_zero_filled_by_the_forward_pass(at):
    """gsr_backward_prefill (include/gsr.h): gradient outputs announced before a forward pass are zero-filled by that pass on the
    library's second stream (`prefill_at` 1: beside the compositing kernel, 2: beside the list-ordering kernel; 0: announcements are
    ignored) and the backward call that names them skips its own fill.  Buffers that held NaNs must hold exact zeros after the forward
    call (or still NaNs with 0), and -- with the deterministic reverse pass -- the gradients must equal, bit for bit, those of a
    backward call that was never announced; so must those of a render whose announcement was overtaken by another forward pass."""
    import torch
    from gaussian_transformer_amd import _lib
    from gaussian_transformer_amd.rasterizer import GaussianRasterizationSettings, get_backend
    be = get_backend()
    sc = synth.make_scene(P=300000, width=800, height=800, sh_degree=2, max_sh_degree=3, s0=0.01, seed=32)
    cam, P = sc.camera, sc.P
    t = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device="cuda")
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(sc.bg), 1.0, t(cam.world_view_transform),
                                       t(cam.full_proj_transform), sc.sh_degree, t(cam.camera_center), False, False)
    e = torch.empty(0, device="cuda")
    a = (t(sc.means3D), t(sc.shs), e, t(sc.opacities.reshape(P, 1)), t(sc.scales), t(sc.rotations), e)
    dL = t(sc.dL_dimage)
    M = int(a[1].shape[1])
    dev = a[0].device
    names = ("means3D", "means2D", "sh", "colors", "opacity", "scales", "rots", "cov3D")

    def backward(f):
        n, color, radii, geom, binning, img = f
        g = be.backward(rs, n, dL, a[0], radii, a[1], a[2], a[4], a[5], a[6], geom, binning, img)
        return {k: (None if v is None else v.cpu().numpy()) for k, v in zip(names, g)}

    def announce():
        grads = be._gradient_outputs(dev, P, M, 0, True, False, False, None)
        for g in grads:
            if g is not None:
                g.fill_(float("nan"))
        g_means3D, g_means2D, g_sh, g_colors, g_opacity, g_scales, g_rots, g_cov3D, _ = grads
        p = lambda x: None if x is None or x.numel() == 0 else x.data_ptr()
        _lib.check(be.lib.gsr_backward_prefill(P, M, p(g_means2D), p(g_opacity), p(g_colors), p(g_means3D), p(g_cov3D), p(g_sh), p(g_scales),
                                               p(g_rots), None), "gsr_backward_prefill")
        return grads

    _lib.set_option("deterministic_bwd", 1); _lib.set_option("dense_pergauss", 1); _lib.set_option("prefill_at", at)
    try:
        ref = backward(be.forward(rs, *a))                         # never announced
        grads = announce()
        f = be.forward(rs, *a)
        torch.cuda.synchronize()
        for g in grads:
            if g is not None and g.numel():
                assert bool(torch.isnan(g).all()) if at == 0 else bool((g == 0).all())
        be._announced[dev.index] = (f[3].data_ptr(), P, M, grads, None)     # what forward(announce_backward=True) leaves behind
        got = backward(f)
        # an announcement overtaken by another forward pass: the first render's backward call fills by itself
        grads1 = announce()
        f1 = be.forward(rs, *a)
        grads2 = announce()
        f2 = be.forward(rs, *a)
        for g in grads1:
            if g is not None:
                g.fill_(float("nan"))
        be._announced[dev.index] = (f1[3].data_ptr(), P, M, grads1, None)
        late = backward(f1)
        del f2, grads2
    finally:
        _lib.set_option("deterministic_bwd", 0); _lib.set_option("dense_pergauss", 2); _lib.set_option("prefill_at", 1)
        be._announced.clear()
    some = False
    for k, v in ref.items():
        if v is None:
            assert got[k] is None and late[k] is None
            continue
        assert np.array_equal(v, got[k]), (k, "announced")
        assert np.array_equal(v, late[k]), (k, "overtaken")
        some = some or np.abs(v).max() > 0
    assert some
