"""CPU: the constructive image certificate of tests/helpers.py (what the -m gpu full-size tests apply to the HIP image).
The numpy re-composite of one pixel follows oracle/gsr_ref.c operation for operation; a pixel that differs from the oracle because
ONE borderline decision went the other way is explained, a pixel that differs for any other reason is not."""
import numpy as np

from gaussian_transformer_amd import synth
from oracle import ref
from tests.helpers import BORDERLINE, certify_image_constructive, oracle_scene, recomposite_pixel


def _scene():
    sc = synth.make_scene(P=6000, width=128, height=96, sh_degree=2, s0=0.04, seed=5, bg=(0.2, 0.3, 0.1))
    S = oracle_scene(sc)
    return S, ref.get("f32").forward(S)


def test_numpy_recomposite_equals_the_c_oracle():
    S, f = _scene()
    st = f["state"]; geom, b = st.geom(), st.binning()
    gridx = (S.W + 15) // 16
    rng = np.random.default_rng(0)
    for _ in range(60):
        x, y = int(rng.integers(S.W)), int(rng.integers(S.H))
        r0, r1 = b["ranges"][(y // 16) * gridx + x // 16]
        c, _ = recomposite_pixel(geom, b["vals"], r0, r1, x, y, S.bg)
        assert np.abs(c - f["color"][:, y, x]).max() < 5e-7


def test_reversed_borderline_decision_is_explained_and_anything_else_is_not():
    S, f = _scene()
    st = f["state"]; geom, b = st.geom(), st.binning()
    gridx = (S.W + 15) // 16
    # a pixel with a borderline decision whose reversal moves it by more than 1e-4
    found = None
    for y in range(S.H):
        for x in range(S.W):
            r0, r1 = b["ranges"][(y // 16) * gridx + x // 16]
            base, dec = recomposite_pixel(geom, b["vals"], r0, r1, x, y, S.bg)
            for p, k, m in dec:
                if m < BORDERLINE:
                    c, _ = recomposite_pixel(geom, b["vals"], r0, r1, x, y, S.bg, flips=[(p, k)])
                    if np.abs(c - base).max() > 2e-4:
                        found = (x, y, c); break
            if found:
                break
        if found:
            break
    assert found is not None, "the scene has no borderline decision that matters (change the seed)"
    x, y, c = found
    img = f["color"].copy()
    img[:, y, x] = c
    st1 = certify_image_constructive(img, f)
    assert st1["over"] == 1 and st1["certified"] == 1 and not st1["unexplained"]
    img[:, (y + 7) % S.H, (x + 9) % S.W] += np.float32(3e-4)          # off by 3e-4 for no reason
    st2 = certify_image_constructive(img, f)
    assert st2["over"] == 2 and st2["certified"] == 1 and len(st2["unexplained"]) == 1
