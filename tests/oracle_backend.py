"""Oracle-backed stand-in for the HIP backend, for CPU tests of the HOST logic only
(validation, autograd plumbing, gloo data parallelism).  Test infrastructure: injected with
rasterizer._set_backend_for_tests(); the product never falls back to it."""
import numpy as np
import torch

from oracle import ref


class OracleBackend:
    name = "oracle"

    def __init__(self, precision="f32"):
        self.r = ref.get(precision)
        self._states = {}
        self._next = 1

    @staticmethod
    def _np(t):
        return None if t is None or t.numel() == 0 else t.detach().cpu().numpy()

    def forward(self, rs, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp):
        S = ref.Scene(W=int(rs.image_width), H=int(rs.image_height), tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                      viewmatrix=self._np(rs.viewmatrix), projmatrix=self._np(rs.projmatrix), campos=self._np(rs.campos),
                      means3D=self._np(means3D) if means3D.numel() else np.zeros((0, 3), np.float32),
                      opacities=self._np(opacities) if opacities.numel() else np.zeros((0,), np.float32),
                      bg=self._np(rs.bg), sh_degree=int(rs.sh_degree), shs=self._np(shs),
                      colors_precomp=self._np(colors_precomp), scales=self._np(scales), rotations=self._np(rotations),
                      cov3D_precomp=self._np(cov3D_precomp), scale_modifier=float(rs.scale_modifier))
        P = means3D.shape[0]
        if P == 0:
            z = torch.zeros((3, rs.image_height, rs.image_width)); e = torch.zeros((0,), dtype=torch.uint8)
            return 0, z, torch.zeros((0,), dtype=torch.int32), e, e, e
        f = self.r.forward(S)
        key = self._next; self._next += 1
        self._states[key] = f
        handle = torch.tensor([key], dtype=torch.int64)
        return (f["num_rendered"], torch.from_numpy(f["color"].astype(np.float32)), torch.from_numpy(f["radii"].copy()),
                handle, torch.zeros((0,), dtype=torch.uint8), torch.zeros((0,), dtype=torch.uint8))

    def backward(self, rs, num_rendered, dL_dpix, means3D, radii, shs, colors_precomp, scales, rotations, cov3D_precomp,
                 geom, binning, img):
        P = means3D.shape[0]
        f = self._states.pop(int(geom[0]))
        g = self.r.backward(f, dL_dpix.detach().cpu().numpy())
        t = lambda a, shape: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).reshape(shape) if a is not None else torch.zeros((0,))
        M = shs.shape[1] if shs.numel() else 0
        return (t(g["dL_dmeans3D"], (P, 3)), t(g["dL_dmeans2D"], (P, 3)), t(g["dL_dsh"], (P, M, 3)) if M else torch.zeros((0,)),
                t(g["dL_dcolors"], (P, 3)), t(g["dL_dopacity"], (P, 1)), t(g["dL_dscales"], (P, 3)), t(g["dL_drots"], (P, 4)),
                t(g["dL_dcov3D"], (P, 6)))

    def mark_visible(self, positions, viewmatrix, projmatrix):
        return torch.from_numpy(self.r.mark_visible(self._np(positions), self._np(viewmatrix)))
