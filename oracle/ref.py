"""ctypes front-end of the CPU restatement (oracle/gsr_ref.c).

TEST INFRASTRUCTURE ONLY -- see the header of gsr_ref.c.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product package.
Parity status: "parity unpinned" at the rasterizer boundary (no reference test pins it,
SURVEY.md 8c); sub-stages are pinned by tests/golden/*.npz.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile the restatement with gcc (both float and double variants)."""
    need = force or not all(
        os.path.exists(os.path.join(_BUILD, n)) for n in ("libgsr_ref_f32.so", "libgsr_ref_f64.so")
    )
    src = os.path.join(_HERE, "gsr_ref.c")
    if not need:
        newest = min(os.path.getmtime(os.path.join(_BUILD, n)) for n in ("libgsr_ref_f32.so", "libgsr_ref_f64.so"))
        need = os.path.getmtime(src) > newest
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def _scene_struct(real):
    class Scene(C.Structure):
        _fields_ = [
            ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
            ("prefiltered", C.c_int32),
            ("scale_modifier", real), ("tanfovx", real), ("tanfovy", real),
            ("bg", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
            ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p),
            ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
            ("campos", C.c_void_p),
        ]
    return Scene


@dataclass
class Scene:
    """Inputs of one render, numpy arrays (any float dtype; converted per precision)."""
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray          # [4,4] memory of world_view_transform
    projmatrix: np.ndarray          # [4,4] memory of full_proj_transform
    campos: np.ndarray              # [3]
    means3D: np.ndarray             # [P,3]
    opacities: np.ndarray           # [P] or [P,1]
    bg: np.ndarray = field(default_factory=lambda: np.zeros(3))
    sh_degree: int = 0
    shs: Optional[np.ndarray] = None             # [P,M,3]
    colors_precomp: Optional[np.ndarray] = None  # [P,3]
    scales: Optional[np.ndarray] = None          # [P,3]
    rotations: Optional[np.ndarray] = None       # [P,4] (r,x,y,z)
    cov3D_precomp: Optional[np.ndarray] = None   # [P,6]
    scale_modifier: float = 1.0
    prefiltered: bool = False


class RefRasterizer:
    """One loaded precision variant of the oracle ('f32' or 'f64')."""

    def __init__(self, precision: str = "f32"):
        assert precision in ("f32", "f64")
        build()
        self.precision = precision
        self.np_real = np.float32 if precision == "f32" else np.float64
        self.c_real = C.c_float if precision == "f32" else C.c_double
        self.lib = C.CDLL(os.path.join(_BUILD, f"libgsr_ref_{precision}.so"))
        self.SceneStruct = _scene_struct(self.c_real)
        L = self.lib
        L.gsr_ref_forward.restype = C.c_void_p
        L.gsr_ref_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.gsr_ref_backward.restype = C.c_int
        L.gsr_ref_backward.argtypes = [C.c_void_p] * 12 + [C.c_int32]
        L.gsr_ref_free.argtypes = [C.c_void_p]
        L.gsr_ref_num_rendered.restype = C.c_int64
        L.gsr_ref_num_rendered.argtypes = [C.c_void_p]
        L.gsr_ref_get_geom.argtypes = [C.c_void_p] * 8
        L.gsr_ref_get_binning.argtypes = [C.c_void_p] * 4
        L.gsr_ref_get_image_state.argtypes = [C.c_void_p] * 3
        L.gsr_ref_get_margin.argtypes = [C.c_void_p] * 2
        L.gsr_ref_set_tile_row_band.argtypes = [C.c_int32, C.c_int32]
        L.gsr_ref_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gsr_ref_max_threads.restype = C.c_int32
        L.gsr_ref_get_timings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        assert L.gsr_ref_real_bytes() == np.dtype(self.np_real).itemsize

    # -- helpers --
    def _arr(self, a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a), dtype=self.np_real)
        if shape is not None:
            a = a.reshape(shape)
        return a

    def _pack(self, sc: Scene):
        P = int(np.asarray(sc.means3D).shape[0])
        keep = dict(
            bg=self._arr(sc.bg, (3,)), means3D=self._arr(sc.means3D, (P, 3)),
            shs=self._arr(sc.shs), colors_precomp=self._arr(sc.colors_precomp),
            opacities=self._arr(sc.opacities, (P,)), scales=self._arr(sc.scales),
            rotations=self._arr(sc.rotations), cov3D_precomp=self._arr(sc.cov3D_precomp),
            viewmatrix=self._arr(sc.viewmatrix, (16,)), projmatrix=self._arr(sc.projmatrix, (16,)),
            campos=self._arr(sc.campos, (3,)),
        )
        if (keep["shs"] is None) == (keep["colors_precomp"] is None):
            raise ValueError("exactly one of shs / colors_precomp")
        if (keep["scales"] is None or keep["rotations"] is None) == (keep["cov3D_precomp"] is None):
            raise ValueError("exactly one of (scales, rotations) / cov3D_precomp")
        M = int(keep["shs"].shape[1]) if keep["shs"] is not None else 0
        s = self.SceneStruct()
        s.P, s.D, s.M, s.W, s.H = P, int(sc.sh_degree), M, int(sc.W), int(sc.H)
        s.prefiltered = int(sc.prefiltered)
        s.scale_modifier, s.tanfovx, s.tanfovy = float(sc.scale_modifier), float(sc.tanfovx), float(sc.tanfovy)
        for k, v in keep.items():
            setattr(s, k, v.ctypes.data if v is not None else None)
        return s, keep, P, M

    # -- API --
    def forward(self, sc: Scene, nthreads: int = 1):
        """Returns dict(color[3,H,W], radii[P], state=<handle>, num_rendered)."""
        s, keep, P, M = self._pack(sc)
        color = np.zeros((3, sc.H, sc.W), dtype=self.np_real)
        radii = np.zeros((max(P, 1),), dtype=np.int32)
        st = self.lib.gsr_ref_forward(C.byref(s), color.ctypes.data, radii.ctypes.data, int(nthreads))
        if not st:
            raise RuntimeError("gsr_ref_forward failed")
        return dict(color=color, radii=radii[:P], state=_State(self, st, s, keep, P, M, sc),
                    num_rendered=int(self.lib.gsr_ref_num_rendered(st)))

    def backward(self, fwd: dict, dL_dpix: np.ndarray, nthreads: int = 1) -> dict:
        st: _State = fwd["state"]
        P, M = st.P, st.M
        r = self.np_real
        n = max(P, 1)
        g = dict(
            dL_dmeans2D=np.zeros((n, 3), r), dL_dconic=np.zeros((n, 4), r), dL_dopacity=np.zeros((n,), r),
            dL_dcolors=np.zeros((n, 3), r), dL_dmeans3D=np.zeros((n, 3), r), dL_dcov3D=np.zeros((n, 6), r),
            dL_dsh=np.zeros((n, M, 3), r) if M > 0 else None,
            dL_dscales=np.zeros((n, 3), r) if st.keep["scales"] is not None else None,
            dL_drots=np.zeros((n, 4), r) if st.keep["scales"] is not None else None,
        )
        dpix = self._arr(dL_dpix, (3, st.sc.H, st.sc.W))
        ptr = lambda a: a.ctypes.data if a is not None else None
        rc = self.lib.gsr_ref_backward(
            C.byref(st.struct), st.handle, dpix.ctypes.data, ptr(g["dL_dmeans2D"]), ptr(g["dL_dconic"]),
            ptr(g["dL_dopacity"]), ptr(g["dL_dcolors"]), ptr(g["dL_dmeans3D"]), ptr(g["dL_dcov3D"]),
            ptr(g["dL_dsh"]), ptr(g["dL_dscales"]), ptr(g["dL_drots"]), int(nthreads))
        if rc != 0:
            raise RuntimeError(f"gsr_ref_backward rc={rc}")
        return {k: (v[:P] if v is not None else None) for k, v in g.items()}

    def mark_visible(self, means3D, viewmatrix):
        m = self._arr(means3D); v = self._arr(viewmatrix, (16,))
        out = np.zeros((m.shape[0],), dtype=np.uint8)
        self.lib.gsr_ref_mark_visible(m.shape[0], m.ctypes.data, v.ctypes.data, out.ctypes.data)
        return out.astype(bool)

    def set_tile_row_band(self, ty0: int = 0, ty1: int = 0) -> None:
        """Timing aid (bench.py single-thread leg): composite only tile rows [ty0, ty1); (0, 0) = whole image."""
        self.lib.gsr_ref_set_tile_row_band(int(ty0), int(ty1))

    def max_threads(self) -> int:
        return int(self.lib.gsr_ref_max_threads())


class _State:
    def __init__(self, owner: RefRasterizer, handle, struct, keep, P, M, sc):
        self.owner, self.handle, self.struct, self.keep, self.P, self.M, self.sc = owner, handle, struct, keep, P, M, sc

    def __del__(self):
        try:
            if self.handle:
                self.owner.lib.gsr_ref_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def geom(self) -> dict:
        r, P = self.owner.np_real, max(self.P, 1)
        d = dict(depth=np.zeros(P, r), xy=np.zeros((P, 2), r), conic_o=np.zeros((P, 4), r), rgb=np.zeros((P, 3), r),
                 cov3D=np.zeros((P, 6), r), tiles_touched=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8))
        if self.P > 0:
            self.owner.lib.gsr_ref_get_geom(self.handle, *[d[k].ctypes.data for k in
                                                           ("depth", "xy", "conic_o", "rgb", "cov3D", "tiles_touched", "clamped")])
        return {k: v[:self.P] for k, v in d.items()}

    def binning(self) -> dict:
        N = int(self.owner.lib.gsr_ref_num_rendered(self.handle))
        T = ((self.sc.W + 15) // 16) * ((self.sc.H + 15) // 16)
        keys = np.zeros(max(N, 1), np.uint64); vals = np.zeros(max(N, 1), np.uint32); ranges = np.zeros((T, 2), np.uint32)
        self.owner.lib.gsr_ref_get_binning(self.handle, keys.ctypes.data, vals.ctypes.data, ranges.ctypes.data)
        return dict(keys=keys[:N], vals=vals[:N], ranges=ranges)

    def timings(self) -> dict:
        """Wall-clock seconds per stage of the last forward / backward on this state."""
        f = (C.c_double * 3)(); b = (C.c_double * 2)()
        self.owner.lib.gsr_ref_get_timings(self.handle, f, b)
        return {"fwd.preprocess": f[0], "fwd.scan+emit+sort+ranges": f[1], "fwd.composite": f[2],
                "bwd.composite": b[0], "bwd.pergauss": b[1]}

    def decision_margin(self) -> np.ndarray:
        """[H,W]: per pixel, the smallest distance of any discrete decision of S9 (power > 0, alpha < 1/255,
        T(1-alpha) < 1e-4) from its threshold -- relative for alpha and T, absolute for power.  A pixel can only
        legitimately differ from another correct implementation by more than rounding if this is tiny."""
        m = np.zeros((self.sc.H, self.sc.W), self.owner.np_real)
        self.owner.lib.gsr_ref_get_margin(self.handle, m.ctypes.data)
        return m

    def image_state(self) -> dict:
        r = self.owner.np_real
        fT = np.zeros((self.sc.H, self.sc.W), r); nc = np.zeros((self.sc.H, self.sc.W), np.uint32)
        self.owner.lib.gsr_ref_get_image_state(self.handle, fT.ctypes.data, nc.ctypes.data)
        return dict(final_T=fT, n_contrib=nc)


_cache = {}


def get(precision: str = "f32") -> RefRasterizer:
    if precision not in _cache:
        _cache[precision] = RefRasterizer(precision)
    return _cache[precision]
