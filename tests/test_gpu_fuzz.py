"""-m gpu: seeded random scenes over the whole default path (depth_order.hip + tile_lists.hip + compositing) against the
oracle: sizes, aspect ratios, splat scales, depth ranges and SH degrees drawn at random; per-tile lists bit-exact under
the upstream tile rule, image within the RGB bar, and the default (exact-culling) mode within the RGB bar too."""
import numpy as np
import pytest

from gaussian_transformer_amd import synth
from oracle import ref
from tests.helpers import GRAD_RTOL, assert_image_close, grad_err, hip_forward_backward, oracle_scene
from tests.test_gpu_parity import _stage_dump

pytestmark = pytest.mark.gpu


def _draw(rng):
    P = int(rng.choice([1, 7, 300, 2000, 9000, 30000]))
    W = int(rng.integers(8, 700)); H = int(rng.integers(8, 500))
    zmin = float(rng.choice([0.05, 0.5, 2.0, 5.0])); zmax = zmin * float(rng.choice([1.0, 1.2, 3.0, 40.0]))
    return dict(P=P, width=W, height=H, sh_degree=int(rng.integers(0, 4)), s0=float(10 ** rng.uniform(-2.5, -0.3)),
                seed=int(rng.integers(1 << 30)), zmin=zmin, zmax=zmax)


@pytest.mark.parametrize("batch", range(6))
def test_random_scenes(batch):
    from gaussian_transformer_amd import _lib
    rng = np.random.default_rng(1234 + batch)
    for _ in range(6):
        kw = _draw(rng)
        sc = synth.make_scene(**kw)
        S = oracle_scene(sc)
        f = ref.get("f32").forward(S)
        ob = f["state"].binning()
        _lib.set_option("exact_tile_cull", 0)
        _lib.set_option("depth_buckets", 2)               # the bucketed order whatever P is
        try:
            h = _stage_dump(S)
            np.testing.assert_array_equal(h["radii"], f["radii"], err_msg=str(kw))
            assert h["n"] == f["num_rendered"], kw
            np.testing.assert_array_equal(h["point_list"], ob["vals"], err_msg=str(kw))
            np.testing.assert_array_equal(h["ranges"], ob["ranges"], err_msg=str(kw))
            assert_image_close(h["color"], f["color"])
            _lib.set_option("exact_tile_cull", 1)
            h2 = _stage_dump(S)
            assert h2["n"] <= f["num_rendered"], kw
            assert_image_close(h2["color"], f["color"])
            if kw["P"] <= 2000:                            # gradients too, in the default mode
                dL = rng.normal(size=(3, S.H, S.W)).astype(np.float32)
                g = ref.get("f32").backward(f, dL)
                hb = hip_forward_backward(S, dL)
                for a, b in (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
                             ("rotations", "dL_drots")):
                    assert grad_err(hb["grads"][a], g[b]) < GRAD_RTOL, (kw, a, grad_err(hb["grads"][a], g[b]))
        finally:
            _lib.set_option("exact_tile_cull", 1)
            _lib.set_option("depth_buckets", 1)
            _lib.set_option("depth_log_map", 0)
