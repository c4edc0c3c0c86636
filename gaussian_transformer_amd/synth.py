"""Seeded synthetic scenes for the BASELINE.json configs (SURVEY.md 8d "Synthetic generator").

The reference ships no trained point_cloud.ply and no camera poses (images.bin missing,
/root/reference/.MISSING_LARGE_BLOBS:1-3), so every config is synthesised: Gaussians sampled
inside the frustum of a camera at the origin looking down +z.  All float32, NumPy
default_rng(seed).
"""
from __future__ import annotations

import math
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


# The five BASELINE.json configs (concrete sizes from SURVEY.md 8d).
CONFIGS = {
    "cfg1_plumbing_10k_256": dict(P=10_000, width=256, height=256, sh_degree=0, s0=0.05),
    "cfg2_table_300k_800": dict(P=300_000, width=800, height=800, sh_degree=3, s0=0.01, tanfovx=0.5),
    "cfg3_synth_1M_1080p": dict(P=1_000_000, width=1920, height=1080, sh_degree=3, s0=0.01),
    "cfg4_tiramisu_303k_1600x900": dict(P=303_570, width=1600, height=900, sh_degree=3, s0=0.01, tanfovx=0.6132),
    "cfg5_stress_5M_4k": dict(P=5_000_000, width=3840, height=2160, sh_degree=3, s0=0.005),
}


def make_config(name: str, seed: int = 0, **over) -> SyntheticScene:
    kw = dict(CONFIGS[name]); kw.update(over)
    return make_scene(seed=seed, **kw)
