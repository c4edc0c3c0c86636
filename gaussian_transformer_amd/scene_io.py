"""Scene-side loaders either side of the rasterizer (second half): COLMAP text models, the camera
records built from a COLMAP model, NeRF++ scene normalisation, the image-resolution rule of the training cameras, the
`cameras.json` entry and the training checkpoint tuple.  Host Python (CPU-side byte work done once per run).

Restates the behaviour (not the code) of the reference:
  scene/colmap_loader.py:83-123     read_points3D_text
  scene/colmap_loader.py:156-178    read_intrinsics_text   (asserts PINHOLE: the rest of the pipeline assumes it)
  scene/colmap_loader.py:244-271    read_extrinsics_text
  scene/dataset_readers.py:43-66    getNerfppNorm          (translate = -mean camera centre, radius = 1.1 * max distance)
  scene/dataset_readers.py:68-105   readColmapCameras      (R = qvec2rotmat(q)^T, FoV from the focal lengths)
  scene/dataset_readers.py:132-177  readColmapSceneInfo    (bin first, text as fall-back; llffhold split; PLY conversion)
  utils/camera_utils.py:19-52       loadCam                (resolution 1/2/4/8 = divisor with round(); -1 = cap at 1600 px wide
                                                            with int(); any other value = target width)
  utils/camera_utils.py:62-82       camera_to_JSON
  scene/gaussian_model.py:61-93     capture / restore ; train.py:130-132 (checkpoint = (capture(), iteration))
Pinned by tests/golden/scene_io.npz + scene_io_*.txt, produced by running the reference's own functions
(oracle/make_golden.py): tests/test_scene_io.py.
"""
from __future__ import annotations

import json
import math
import os
from collections import namedtuple
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import io as gio
from .camera import focal2fov, fov2focal, make_camera, world_to_view

CameraInfo = namedtuple("CameraInfo", ["uid", "R", "T", "FovY", "FovX", "image", "image_path", "image_name", "width", "height"])
SceneInfo = namedtuple("SceneInfo", ["point_cloud", "train_cameras", "test_cameras", "nerf_normalization", "ply_path"])


def _data_lines(path: str):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and line[0] != "#":
                yield line


def read_points3D_text(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """points3D.txt -> (xyz [N,3], rgb [N,3] as 0..255 floats, error [N,1]); the track (elements 8..) is ignored."""
    rows = [ln.split() for ln in _data_lines(path)]
    n = len(rows)
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    for i, e in enumerate(rows):
        xyz[i] = [float(v) for v in e[1:4]]
        rgb[i] = [int(v) for v in e[4:7]]
        err[i] = float(e[7])
    return xyz, rgb, err


def read_cameras_text(path: str) -> Dict[int, gio.ColmapCamera]:
    """cameras.txt.  Like the reference, only PINHOLE is accepted in the text form."""
    out = {}
    for ln in _data_lines(path):
        e = ln.split()
        if e[1] != "PINHOLE":
            raise AssertionError("While the loader support other types, the rest of the code assumes PINHOLE")
        out[int(e[0])] = gio.ColmapCamera(id=int(e[0]), model=e[1], width=int(e[2]), height=int(e[3]),
                                          params=np.array([float(v) for v in e[4:]]))
    return out


def read_images_text(path: str) -> Dict[int, gio.ColmapImage]:
    """images.txt: two lines per image (pose line, then x y point3D_id triples, possibly empty)."""
    out = {}
    with open(path, "r") as f:
        while True:
            line = f.readline()
            if not line:
                break
            line = line.strip()
            if not line or line[0] == "#":
                continue
            e = line.split()
            pts = f.readline().split()
            xys = np.column_stack([[float(v) for v in pts[0::3]], [float(v) for v in pts[1::3]]]) if pts else np.zeros((0, 2))
            out[int(e[0])] = gio.ColmapImage(id=int(e[0]), qvec=np.array([float(v) for v in e[1:5]]),
                                             tvec=np.array([float(v) for v in e[5:8]]), camera_id=int(e[8]), name=e[9],
                                             xys=xys, point3D_ids=np.array([int(v) for v in pts[2::3]], dtype=np.int64))
    return out


def colmap_camera_infos(extrinsics: Dict[int, gio.ColmapImage], intrinsics: Dict[int, gio.ColmapCamera], images_folder: str,
                        open_image=None) -> List[CameraInfo]:
    """One CameraInfo per registered image.  `open_image(path)` defaults to PIL.Image.open when PIL is importable and the
    file exists, else the record carries image=None (width / height always come from the intrinsics)."""
    infos = []
    for key in extrinsics:
        extr = extrinsics[key]
        intr = intrinsics[extr.camera_id]
        R = np.transpose(gio.qvec2rotmat(extr.qvec))
        T = np.array(extr.tvec)
        if intr.model == "SIMPLE_PINHOLE":
            fy = fx = intr.params[0]
        elif intr.model == "PINHOLE":
            fx, fy = intr.params[0], intr.params[1]
        else:
            raise AssertionError("Colmap camera model not handled: only undistorted datasets (PINHOLE or SIMPLE_PINHOLE cameras) supported!")
        image_path = os.path.join(images_folder, os.path.basename(extr.name))
        image = None
        if open_image is not None:
            image = open_image(image_path)
        elif os.path.exists(image_path):
            try:
                from PIL import Image
                image = Image.open(image_path)
            except ImportError:
                image = None
        infos.append(CameraInfo(uid=intr.id, R=R, T=T, FovY=focal2fov(fy, intr.height), FovX=focal2fov(fx, intr.width), image=image,
                                image_path=image_path, image_name=os.path.basename(image_path).split(".")[0],
                                width=intr.width, height=intr.height))
    return infos


def nerfpp_norm(cam_infos: Sequence) -> dict:
    """{"translate": -mean camera centre, "radius": 1.1 * largest distance of a camera from that mean}."""
    centres = []
    for c in cam_infos:
        W2C = world_to_view(c.R, c.T)                       # float32, as getWorld2View2 returns
        centres.append(np.linalg.inv(W2C)[:3, 3:4])
    centres = np.hstack(centres)
    centre = np.mean(centres, axis=1, keepdims=True)
    diagonal = np.max(np.linalg.norm(centres - centre, axis=0, keepdims=True))
    return {"translate": -centre.flatten(), "radius": diagonal * 1.1}


def read_colmap_scene(path: str, images: Optional[str] = None, eval: bool = False, llffhold: int = 8, open_image=None) -> SceneInfo:
    """sparse/0/{images,cameras}.bin, falling back to the .txt pair; cameras sorted by image name; every llffhold-th one
    held out when eval; points3D.ply written from points3D.bin / .txt the first time."""
    sp = os.path.join(path, "sparse/0")
    try:
        extr = gio.read_images_binary(os.path.join(sp, "images.bin"))
        intr = gio.read_cameras_binary(os.path.join(sp, "cameras.bin"))
    except Exception:
        extr = read_images_text(os.path.join(sp, "images.txt"))
        intr = read_cameras_text(os.path.join(sp, "cameras.txt"))
    infos = sorted(colmap_camera_infos(extr, intr, os.path.join(path, "images" if images is None else images), open_image),
                   key=lambda c: c.image_name)
    if eval:
        train = [c for i, c in enumerate(infos) if i % llffhold != 0]
        test = [c for i, c in enumerate(infos) if i % llffhold == 0]
    else:
        train, test = infos, []
    ply_path = os.path.join(sp, "points3D.ply")
    if not os.path.exists(ply_path):
        try:
            xyz, rgb, _ = gio.read_points3D_binary(os.path.join(sp, "points3D.bin"))
        except Exception:
            xyz, rgb, _ = read_points3D_text(os.path.join(sp, "points3D.txt"))
        gio.store_point_cloud(ply_path, xyz, rgb)
    try:
        pcd = gio.fetch_point_cloud(ply_path)
    except Exception:
        pcd = None
    return SceneInfo(point_cloud=pcd, train_cameras=train, test_cameras=test, nerf_normalization=nerfpp_norm(train), ply_path=ply_path)


def training_resolution(orig_w: int, orig_h: int, resolution, resolution_scale: float = 1.0) -> Tuple[int, int]:
    """(width, height) a training camera's image is resized to.  resolution in {1, 2, 4, 8}: that divisor (times
    resolution_scale), rounded; -1: images wider than 1600 px are brought down to 1600 wide; anything else: the target width.
    The last two truncate with int()."""
    if resolution in (1, 2, 4, 8):
        return round(orig_w / (resolution_scale * resolution)), round(orig_h / (resolution_scale * resolution))
    if resolution == -1:
        global_down = orig_w / 1600 if orig_w > 1600 else 1
    else:
        global_down = orig_w / resolution
    scale = float(global_down) * float(resolution_scale)
    return int(orig_w / scale), int(orig_h / scale)


def load_camera(cam_info, uid: int, resolution=-1, resolution_scale: float = 1.0):
    """The rasterizer-facing part of loadCam: the camera matrices at the training resolution (the resized ground-truth
    image stays with the caller).  Returns (CameraMatrices, (width, height))."""
    ow, oh = (cam_info.image.size if getattr(cam_info, "image", None) is not None else (cam_info.width, cam_info.height))
    w, h = training_resolution(ow, oh, resolution, resolution_scale)
    return make_camera(cam_info.R, cam_info.T, cam_info.FovX, cam_info.FovY, w, h), (w, h)


def camera_to_json(uid: int, R, T, FovX: float, FovY: float, width: int, height: int, image_name: str) -> dict:
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = np.asarray(T)
    Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    return {"id": uid, "img_name": image_name, "width": width, "height": height, "position": W2C[:3, 3].tolist(),
            "rotation": [x.tolist() for x in W2C[:3, :3]], "fy": fov2focal(FovY, height), "fx": fov2focal(FovX, width)}


def write_cameras_json(path: str, entries: Sequence[dict]) -> None:
    with open(path, "w") as f:
        json.dump(list(entries), f)


# ---- training checkpoint (scene/gaussian_model.py:61-93, train.py:130-132) -------------------------------------------
CAPTURE_FIELDS = ("active_sh_degree", "_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "max_radii2D",
                  "xyz_gradient_accum", "denom", "optimizer_state_dict", "spatial_lr_scale")


def capture(controller) -> tuple:
    """The 12-tuple GaussianModel.capture() returns, from a densify.DensityController (which owns the optimiser)."""
    m = controller.model
    return (m.active_sh_degree, m._xyz, m._features_dc, m._features_rest, m._scaling, m._rotation, m._opacity, m.max_radii2D,
            m.xyz_gradient_accum, m.denom, controller.optimizer.state_dict(), controller.spatial_lr_scale)


def restore(model_args: tuple, opt=None, **controller_kw):
    """GaussianModel.restore(): parameters from the tuple, a fresh training_setup, then statistics and optimiser state."""
    import torch
    from .densify import DensityController
    from .model import GaussianParams
    (active, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, grad_accum, denom, opt_dict, spatial_lr_scale) = model_args
    M = f_dc.shape[1] + f_rest.shape[1]
    model = GaussianParams(int(round(math.sqrt(M))) - 1)
    model.active_sh_degree = active
    model._xyz, model._features_dc, model._features_rest = xyz, f_dc, f_rest
    model._scaling, model._rotation, model._opacity = scaling, rotation, opacity
    ctl = DensityController(model, opt, spatial_lr_scale=spatial_lr_scale, **controller_kw)
    model.max_radii2D = max_radii2D
    model.xyz_gradient_accum = grad_accum
    model.denom = denom
    ctl.optimizer.load_state_dict(opt_dict)
    return ctl


def save_checkpoint(path: str, controller, iteration: int) -> None:
    """torch.save((capture(), iteration), path) -- train.py:130-132."""
    import torch
    torch.save((capture(controller), iteration), path)


def load_checkpoint(path: str, opt=None, **controller_kw):
    """(controller, first_iter) from a checkpoint this module (or the reference) wrote -- train.py:37-39.  The tuple holds
    tensors, ints, floats and the optimiser's plain dict: loaded with weights_only=True."""
    import torch
    model_args, first_iter = torch.load(path, weights_only=True)
    return restore(model_args, opt, **controller_kw), first_iter
