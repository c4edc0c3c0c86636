"""On-disk formats either side of the rasterizer, without `plyfile`.

Restates the schemas (not the code) of the reference:
  * point-cloud PLY      scene/dataset_readers.py:107-130   x y z nx ny nz (f4) red green blue (u1)
  * Gaussian PLY         scene/gaussian_model.py:177-256     x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* (f4);
                                                             f_dc / f_rest are stored CHANNEL-major
                                                             (transpose(1,2).flatten), :196-197, :229-238
  * COLMAP binary model  scene/colmap_loader.py:125-154 (points3D.bin), :204-241 (cameras.bin), :170-201 (images.bin)
Pinned by tests/golden/io_*.{bin,ply,npz}: bytes cut from the reference's shipped table_ds model and what the
reference's own colmap_loader returns for them (oracle/make_golden.py).
"""
from __future__ import annotations

import struct
from collections import namedtuple
from typing import Dict, Tuple

import numpy as np

_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def read_ply_vertices(path: str) -> np.ndarray:
    """The `vertex` element of a PLY file (ascii or binary little/big endian) as a structured array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_other_before = None, None, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    seen_other_before = True
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None or fmt is None:
            raise ValueError(f"{path}: no vertex element")
        if seen_other_before:
            raise ValueError(f"{path}: elements before `vertex` are not supported")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for j, (n, t) in enumerate(props):
                out[n] = np.array([r[j] for r in rows], dtype=np.float64).astype(t)
            return out
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, end + t) for n, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return data.astype(dt.newbyteorder("=")) if end == ">" else data.copy()


def write_ply_vertices(path: str, vertices: np.ndarray) -> None:
    """Binary little-endian PLY with a single `vertex` element (what plyfile writes for the reference)."""
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {vertices.shape[0]}"]
    fields = []
    for name in vertices.dtype.names:
        t = vertices.dtype[name].str[1:]
        header.append(f"property {_PLY_NAMES[t]} {name}")
        fields.append((name, "<" + t))
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(vertices.astype(np.dtype(fields))).tobytes())


BasicPointCloud = namedtuple("BasicPointCloud", ["points", "colors", "normals"])      # utils/graphics_utils.py:17-20


def fetch_point_cloud(path: str) -> BasicPointCloud:
    """scene/dataset_readers.py:107-113: positions, colours / 255, normals."""
    v = read_ply_vertices(path)
    pos = np.ascontiguousarray(np.vstack([v["x"], v["y"], v["z"]]).T)
    col = np.ascontiguousarray(np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0)
    nrm = np.ascontiguousarray(np.vstack([v["nx"], v["ny"], v["nz"]]).T)
    return BasicPointCloud(points=pos, colors=col, normals=nrm)


def store_point_cloud(path: str, xyz: np.ndarray, rgb: np.ndarray) -> None:
    """scene/dataset_readers.py:115-130."""
    n = xyz.shape[0]
    el = np.zeros(n, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                            ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    el["x"], el["y"], el["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    el["red"], el["green"], el["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    write_ply_vertices(path, el)


def gaussian_attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4):
    """scene/gaussian_model.py:177-189."""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
            + ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def save_gaussians(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    """Raw (pre-activation) parameters -> Gaussian PLY (scene/gaussian_model.py:191-208).
    features_dc [P,1,3], features_rest [P,M-1,3] (coefficient-major, as the model holds them)."""
    a = lambda t: np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float32)
    xyz, dc, rest, op, sc, rot = a(xyz), a(features_dc), a(features_rest), a(opacity), a(scaling), a(rotation)
    P = xyz.shape[0]
    f_dc = np.transpose(dc, (0, 2, 1)).reshape(P, -1)          # channel-major on disk
    f_rest = np.transpose(rest, (0, 2, 1)).reshape(P, -1)
    names = gaussian_attribute_names(f_dc.shape[1], f_rest.shape[1], sc.shape[1], rot.shape[1])
    cols = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, op.reshape(P, 1), sc, rot], axis=1)
    el = np.empty(P, dtype=[(n, "f4") for n in names])
    for j, n in enumerate(names):
        el[n] = cols[:, j]
    write_ply_vertices(path, el)


def load_gaussians(path: str, max_sh_degree: int) -> Dict[str, np.ndarray]:
    """Gaussian PLY -> raw parameters in the model's layout (scene/gaussian_model.py:215-256)."""
    v = read_ply_vertices(path)
    P = v.shape[0]
    by_index = lambda prefix: sorted([n for n in v.dtype.names if n.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).astype(np.float32).reshape(P, 3, 1)
    rest_names = by_index("f_rest_")
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* properties, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    rest = np.stack([v[n] for n in rest_names], axis=1).astype(np.float32).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) \
        if rest_names else np.zeros((P, 3, 0), np.float32)
    scales = np.stack([v[n] for n in by_index("scale_")], axis=1).astype(np.float32)
    rots = np.stack([v[n] for n in by_index("rot")], axis=1).astype(np.float32)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(np.transpose(dc, (0, 2, 1))),
                features_rest=np.ascontiguousarray(np.transpose(rest, (0, 2, 1))),
                opacity=np.asarray(v["opacity"], dtype=np.float32)[:, None], scaling=scales, rotation=rots)


# ---- COLMAP binary model ----
ColmapCamera = namedtuple("ColmapCamera", ["id", "model", "width", "height", "params"])
ColmapImage = namedtuple("ColmapImage", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])
_CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                  5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                  9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}


def read_cameras_binary(path: str) -> Dict[int, ColmapCamera]:
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, model_id, w, h = struct.unpack("<iiQQ", f.read(24))
            name, npar = _CAMERA_MODELS[model_id]
            params = np.array(struct.unpack("<" + "d" * npar, f.read(8 * npar)))
            out[cid] = ColmapCamera(id=cid, model=name, width=w, height=h, params=params)
    return out


def read_points3D_binary(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(xyz [N,3] f64, rgb [N,3] f64 0..255, error [N,1]) like scene/colmap_loader.py:125-154."""
    with open(path, "rb") as f:
        buf = f.read()
    (n,) = struct.unpack_from("<Q", buf, 0)
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    off = 8
    for i in range(n):
        _pid, x, y, z, r, g, b, e = struct.unpack_from("<QdddBBBd", buf, off)
        off += 43
        (track,) = struct.unpack_from("<Q", buf, off)
        off += 8 + 8 * track
        xyz[i] = (x, y, z); rgb[i] = (r, g, b); err[i] = e
    return xyz, rgb, err


def read_images_binary(path: str) -> Dict[int, ColmapImage]:
    """images.bin (poses).  The reference's datasets ship WITHOUT this file (.MISSING_LARGE_BLOBS:1-3), so
    this reader is checked only by a write/read round trip."""
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            iid, = struct.unpack("<i", f.read(4))
            q = np.array(struct.unpack("<dddd", f.read(32))); t = np.array(struct.unpack("<ddd", f.read(24)))
            (cam,) = struct.unpack("<i", f.read(4))
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00" or not c:
                    break
                name += c
            (m,) = struct.unpack("<Q", f.read(8))
            rec = np.frombuffer(f.read(24 * m), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            out[iid] = ColmapImage(id=iid, qvec=q, tvec=t, camera_id=cam, name=name.decode("utf-8"),
                                   xys=np.stack([rec["x"], rec["y"]], 1) if m else np.zeros((0, 2)), point3D_ids=rec["id"].copy())
    return out


def qvec2rotmat(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])
