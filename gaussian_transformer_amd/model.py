"""Minimal parameter container producing the rasterizer's inputs.

Restates only the part of scene/gaussian_model.py that sits on the hot path: the raw
parameters, their activations (exp / sigmoid / L2-normalize, :33-41), the getters (:95-118)
and the densification statistic that consumes the means2D gradient (:405-407).
"Next" rows: initialisation from an SfM cloud (create_from_pcd, :124-146, using this repo's distCUDA2) and the
Gaussian PLY checkpoint (save_ply / load_ply, :191-256, through io.py).  Densification and Adam groups are not here.
"""
from __future__ import annotations

import torch

from .synth import SyntheticScene


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:     # utils/general_utils.py:21-22
    return torch.log(x / (1 - x))


class GaussianParams:
    def __init__(self, sh_degree: int):
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        self._xyz = self._features_dc = self._features_rest = None
        self._scaling = self._rotation = self._opacity = None
        self.xyz_gradient_accum = None
        self.denom = None

    @classmethod
    def from_synthetic(cls, sc: SyntheticScene, device, requires_grad: bool = True, max_sh_degree=None):
        M = sc.shs.shape[1]
        deg = int(round(M ** 0.5)) - 1
        self = cls(deg if max_sh_degree is None else max_sh_degree)
        self.active_sh_degree = sc.sh_degree
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device).requires_grad_(requires_grad)
        self._xyz = t(sc.means3D)
        self._features_dc = t(sc.shs[:, :1, :])
        self._features_rest = t(sc.shs[:, 1:, :])
        self._scaling = torch.log(torch.tensor(sc.scales, dtype=torch.float32, device=device)).requires_grad_(requires_grad)
        self._rotation = t(sc.rotations)
        self._opacity = inverse_sigmoid(torch.tensor(sc.opacities, dtype=torch.float32, device=device)).requires_grad_(requires_grad)
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        return self

    @classmethod
    def create_from_pcd(cls, pcd, sh_degree: int, device, dist2_fn=None):
        """scene/gaussian_model.py:124-146.  `pcd` has .points / .colors (0..1).  `dist2_fn` defaults to this repo's
        simple_knn.distCUDA2 equivalent (HIP; no CPU fallback) and is a parameter only so CPU tests can inject one."""
        import numpy as np
        if dist2_fn is None:
            from simple_knn._C import distCUDA2 as dist2_fn
        self = cls(sh_degree)
        self.active_sh_degree = 0                                       # :48; raised by oneupSHdegree every 1000 its
        pts = torch.tensor(np.asarray(pcd.points), dtype=torch.float32, device=device)
        col = torch.tensor(np.asarray(pcd.colors), dtype=torch.float32, device=device)
        P, M = pts.shape[0], (sh_degree + 1) ** 2
        self._xyz = pts.requires_grad_(True)
        self._features_dc = ((col - 0.5) / 0.28209479177387814)[:, None, :].contiguous().requires_grad_(True)   # RGB2SH, utils/sh_utils.py:114-115
        self._features_rest = torch.zeros((P, M - 1, 3), device=device).requires_grad_(True)
        dist2 = torch.clamp_min(dist2_fn(pts.detach()), 1e-7)
        self._scaling = torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3).requires_grad_(True)
        rots = torch.zeros((P, 4), device=device); rots[:, 0] = 1
        self._rotation = rots.requires_grad_(True)
        self._opacity = inverse_sigmoid(0.1 * torch.ones((P, 1), device=device)).requires_grad_(True)
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        return self

    def oneupSHdegree(self):                                            # :120-122
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def save_ply(self, path: str) -> None:                              # :191-208
        from . import io as _io
        _io.save_gaussians(path, self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)

    @classmethod
    def load_ply(cls, path: str, sh_degree: int, device, requires_grad: bool = True):      # :215-256
        from . import io as _io
        d = _io.load_gaussians(path, sh_degree)
        self = cls(sh_degree)
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device).requires_grad_(requires_grad)
        self._xyz, self._features_dc, self._features_rest = t(d["xyz"]), t(d["features_dc"]), t(d["features_rest"])
        self._opacity, self._scaling, self._rotation = t(d["opacity"]), t(d["scaling"]), t(d["rotation"])
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=device)
        self.denom = torch.zeros((P, 1), device=device)
        return self

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier: float = 1.0):
        """Python cov3D path (scene/gaussian_model.py:27-31, utils/general_utils.py:64-110)."""
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1
