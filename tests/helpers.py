"""Shared test helpers: scene -> oracle Scene conversion, tolerances."""
import numpy as np

from gaussian_transformer_amd import synth
from oracle import ref

# north_star tolerances (BASELINE.json): RGB 1e-4 abs, gradients 1e-3 (relative to the
# largest reference magnitude of that tensor, with the same absolute floor).
RGB_ATOL = 1e-4
GRAD_RTOL = 1e-3


def oracle_scene(sc: synth.SyntheticScene, **over) -> ref.Scene:
    cam = sc.camera
    kw = dict(W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
              viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
              means3D=sc.means3D, opacities=sc.opacities, bg=sc.bg, sh_degree=sc.sh_degree, shs=sc.shs,
              scales=sc.scales, rotations=sc.rotations)
    kw.update(over)
    return ref.Scene(**kw)


def grad_err(a, b):
    """max |a-b| / max(|b|_inf, tiny) -- the 1e-3 criterion."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
