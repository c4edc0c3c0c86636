"""Seeded synthetic scenes for the BASELINE.json configs (SURVEY.md 8d "Synthetic generator").

The reference ships no trained point_cloud.ply and no camera poses (images.bin missing,
/root/reference/.MISSING_LARGE_BLOBS:1-3), so every config is synthesised: Gaussians sampled
inside the frustum of a camera at the origin looking down +z.  All float32, NumPy
default_rng(seed).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .camera import CameraMatrices, focal2fov, fov2focal, make_camera


@dataclass
class SyntheticScene:
    camera: CameraMatrices
    means3D: np.ndarray      # [P,3]
    scales: np.ndarray       # [P,3]  (already exp-activated)
    rotations: np.ndarray    # [P,4]  (r,x,y,z), unit norm
    opacities: np.ndarray    # [P,1]  (already sigmoid-activated)
    shs: np.ndarray          # [P,M,3]
    sh_degree: int
    bg: np.ndarray           # [3]
    dL_dimage: np.ndarray    # [3,H,W] upstream gradient, fixed by seed+1

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])


def identity_camera(width: int, height: int, tanfovx: float = math.tan(math.radians(30.0))) -> CameraMatrices:
    """Camera at the origin looking down +z, 60 degree horizontal FoV by default."""
    fovx = 2.0 * math.atan(tanfovx)
    fovy = focal2fov(fov2focal(fovx, width), height)          # tanfovy = tanfovx * H / W
    return make_camera(np.eye(3), np.zeros(3), fovx, fovy, width, height)


def make_scene(P: int, width: int, height: int, sh_degree: int = 3, s0: float = 0.01, seed: int = 0,
               max_sh_degree: Optional[int] = None, zmin: float = 2.0, zmax: float = 10.0,
               tanfovx: float = math.tan(math.radians(30.0)), bg=(0.0, 0.0, 0.0)) -> SyntheticScene:
    rng = np.random.default_rng(seed)
    cam = identity_camera(width, height, tanfovx)
    M = ((max_sh_degree if max_sh_degree is not None else sh_degree) + 1) ** 2
    z = rng.uniform(zmin, zmax, P)
    u = rng.uniform(-1.0, 1.0, P)
    v = rng.uniform(-1.0, 1.0, P)
    means = np.stack([u * z * cam.tanfovx, v * z * cam.tanfovy, z], axis=1).astype(np.float32)
    scales = np.exp(rng.normal(math.log(s0), 0.6, (P, 3))).astype(np.float32)
    q = rng.normal(0.0, 1.0, (P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, (P, 1))))).astype(np.float32)
    shs = rng.normal(0.0, 0.1, (P, M, 3))
    shs[:, 0, :] = rng.normal(0.0, 0.5, (P, 3))
    rng_g = np.random.default_rng(seed + 1)
    dL = (rng_g.normal(0.0, 1.0, (3, height, width)) / (3.0 * height * width)).astype(np.float32)
    return SyntheticScene(cam, means, scales, q.astype(np.float32), opac, shs.astype(np.float32), sh_degree,
                          np.asarray(bg, dtype=np.float32), dL)


# ---- dataset-derived scenes -------------------------------------------------
# The reference ships the SfM clouds of table_ds and tiramisu_ds but no trained point_cloud.ply and no poses
# (images.bin missing), so SURVEY 8d prescribes: every SfM point replicated `copies` times with
# N(0, (0.5 * local scale)^2) jitter, local scale = sqrt(mean squared distance to the 3 nearest points) as
# scene/gaussian_model.py:134-135 computes it, Gaussian scale = 0.5 * local scale, colours from the PLY through
# RGB2SH (utils/sh_utils.py:114-115), higher SH bands N(0, 0.05^2), degree 3.  Opacity / rotation are not prescribed:
# the generic generator's distributions are used.  The clouds are data fixtures under tests/golden/ (copied from
# the reference's datasets by oracle/make_golden.py).
_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SH_C0 = 0.28209479177387814


def _cloud_gaussians(ply_name: str, copies: int, seed: int):
    from scipy.spatial import cKDTree      # host-side scene synthesis only (never on the render path)
    from . import io as gio
    pc = gio.fetch_point_cloud(os.path.join(_GOLDEN, ply_name))
    xyz0 = pc.points.astype(np.float64)
    d, _ = cKDTree(xyz0).query(xyz0, k=4)
    local = np.sqrt(np.maximum((d[:, 1:] ** 2).mean(1), 1e-7))          # scene/gaussian_model.py:134 (clamp_min 1e-7)
    rng = np.random.default_rng(seed)
    n0 = xyz0.shape[0]
    P = n0 * copies
    src = np.repeat(np.arange(n0), copies)
    means = (xyz0[src] + rng.normal(0.0, 1.0, (P, 3)) * (0.5 * local[src])[:, None]).astype(np.float32)
    scales = np.repeat((0.5 * local[src])[:, None], 3, axis=1).astype(np.float32)
    q = rng.normal(0.0, 1.0, (P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 2.0, (P, 1))))).astype(np.float32)
    shs = rng.normal(0.0, 0.05, (P, 16, 3))
    shs[:, 0, :] = (pc.colors[src] - 0.5) / SH_C0                          # RGB2SH
    return means, scales, q.astype(np.float32), opac, shs.astype(np.float32), xyz0


def _upstream_gradient(width: int, height: int, seed: int) -> np.ndarray:
    return (np.random.default_rng(seed + 1).normal(0.0, 1.0, (3, height, width)) / (3.0 * height * width)).astype(np.float32)


def make_table_scene(seed: int = 0, copies: int = 17, width: int = 800, height: int = 800) -> SyntheticScene:
    """BASELINE config 2: table_ds cloud x 17 jitter copies = 299 506 Gaussians, 800 x 800, camera at the origin of the
    COLMAP frame looking down +z (the cloud sits at z = 4..8 there), tanfovx = tanfovy = 0.5."""
    means, scales, q, opac, shs, _ = _cloud_gaussians("table_points3D.ply", copies, seed)
    cam = identity_camera(width, height, tanfovx=0.5)
    return SyntheticScene(cam, means, scales, q, opac, shs, 3, np.zeros(3, np.float32), _upstream_gradient(width, height, seed))


TIRAMISU_FOCAL_4032 = 3287.4641158882314     # tiramisu_ds/sparse/0/cameras.bin (tests/golden/io_colmap.npz cam2_params[0])


def tiramisu_ring_cameras(n: int = 8, width: int = 1600, height: int = 900, centre=None, radius=None):
    """The n ring cameras of BASELINE config 4 (poses are missing from the snapshot): on a circle around the cloud's
    centroid, in the plane perpendicular to COLMAP's vertical (y), at the distance of the COLMAP origin from the
    centroid (where the capture's real cameras roughly stood), looking at the centroid; 1600 x 900 is what
    utils/camera_utils.py:25-39 makes of 4032 x 2268, tanfovx = 2016 / f = 0.6132."""
    from .camera import look_at_camera
    c = np.array([0.24, -0.14, 2.31]) if centre is None else np.asarray(centre, dtype=np.float64)
    r = float(np.linalg.norm(c)) if radius is None else float(radius)
    fovx = focal2fov(TIRAMISU_FOCAL_4032, 4032)
    cams = []
    for k in range(n):
        th = 2.0 * math.pi * k / n
        eye = c + r * np.array([math.sin(th), 0.0, -math.cos(th)])
        cams.append(look_at_camera(eye, c, (0.0, -1.0, 0.0), fovx, width, height))
    return cams


def make_tiramisu_scene(seed: int = 0, copies: int = 9, camera: int = 0, width: int = 1600, height: int = 900) -> SyntheticScene:
    """BASELINE config 4: tiramisu_ds cloud x 9 jitter copies = 303 570 Gaussians seen by ring camera `camera` (0..7)."""
    means, scales, q, opac, shs, xyz0 = _cloud_gaussians("tiramisu_points3D.ply", copies, seed)
    cam = tiramisu_ring_cameras(8, width, height, centre=xyz0.mean(0))[camera]
    return SyntheticScene(cam, means, scales, q, opac, shs, 3, np.zeros(3, np.float32), _upstream_gradient(width, height, seed + 10 * camera))


# The five BASELINE.json configs (concrete sizes from SURVEY.md 8d).
CONFIGS = {
    "cfg1_plumbing_10k_256": dict(P=10_000, width=256, height=256, sh_degree=0, s0=0.05),
    "cfg2_table_300k_800": make_table_scene,
    "cfg3_synth_1M_1080p": dict(P=1_000_000, width=1920, height=1080, sh_degree=3, s0=0.01),
    "cfg4_tiramisu_303k_1600x900": make_tiramisu_scene,
    "cfg5_stress_5M_4k": dict(P=5_000_000, width=3840, height=2160, sh_degree=3, s0=0.005),
    # the generic generator at config 2's / config 4's sizes (round-1 stand-ins, kept as extra parity cases)
    "generic_300k_800": dict(P=300_000, width=800, height=800, sh_degree=3, s0=0.01, tanfovx=0.5),
    "generic_303k_1600x900": dict(P=303_570, width=1600, height=900, sh_degree=3, s0=0.01, tanfovx=0.6132),
}


def make_config(name: str, seed: int = 0, **over) -> SyntheticScene:
    c = CONFIGS[name]
    if callable(c):
        return c(seed=seed, **over)
    kw = dict(c); kw.update(over)
    return make_scene(seed=seed, **kw)
