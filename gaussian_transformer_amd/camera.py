"""Camera matrices in the memory layout the rasterizer reads.

Restates (does not import) the reference's conventions:
  utils/graphics_utils.py:38-49  getWorld2View2
  utils/graphics_utils.py:51-71  getProjectionMatrix (z_sign = +1, znear/zfar mapping)
  utils/graphics_utils.py:73-77  fov2focal / focal2fov
  scene/cameras.py:48-57         zfar=100, znear=0.01; tensors are the TRANSPOSES of the
                                 column-vector matrices, stored row-major, so the kernel reads
                                 x' = m[0]x + m[4]y + m[8]z + m[12].
Pinned by tests/golden/camera.npz (generated from the reference's own functions).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

ZNEAR = 0.01   # scene/cameras.py:49
ZFAR = 100.0   # scene/cameras.py:48


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """4x4 world->camera matrix (column-vector convention), float32.

    R is the camera-to-world rotation as the reference stores it (it is transposed here),
    t the world->camera translation.  translate/scale recentre the camera in world space.
    """
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """4x4 perspective matrix (column-vector convention), float32, w_clip = +z_view."""
    ty, tx = math.tan(fovy / 2.0), math.tan(fovx / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class CameraMatrices:
    """What GaussianRasterizationSettings needs from a camera (numpy float32)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray   # [4,4] = W2C^T
    projection_matrix: np.ndarray      # [4,4] = P^T
    full_proj_transform: np.ndarray    # [4,4] = W2C^T @ P^T
    camera_center: np.ndarray          # [3]

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)


def make_camera(R, t, fovx: float, fovy: float, width: int, height: int,
                translate=(0.0, 0.0, 0.0), scale: float = 1.0,
                znear: float = ZNEAR, zfar: float = ZFAR) -> CameraMatrices:
    """Builds the three tensors exactly as scene/cameras.py:54-57 does (float32 products)."""
    wvt = world_to_view(R, t, translate, scale).T.copy()
    proj = projection_matrix(znear, zfar, fovx, fovy).T.copy()
    full = (wvt.astype(np.float32) @ proj.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(wvt.astype(np.float32))[3, :3].astype(np.float32)
    return CameraMatrices(int(width), int(height), float(fovx), float(fovy), wvt, proj, full, center)


def look_at_camera(eye, target, up, fovx: float, width: int, height: int) -> CameraMatrices:
    """Synthetic pose helper (no reference counterpart: images.bin is missing).

    Camera looks down +z with +y down in image space (COLMAP convention), square pixels.
    """
    eye = np.asarray(eye, dtype=np.float64); target = np.asarray(target, dtype=np.float64)
    fwd = target - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64)); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R_c2w = np.stack([right, down, fwd], axis=1)          # columns = camera axes in world
    t = -R_c2w.T @ eye
    fovy = focal2fov(fov2focal(fovx, width), height)
    return make_camera(R_c2w, t, fovx, fovy, width, height)
