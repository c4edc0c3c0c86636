"""Generates tests/golden/*.npz by IMPORTING the reference's own Python modules on CPU.

Run only in the build container (needs /root/reference); the fixtures it writes are data
(inputs + expected outputs) and are committed, the reference itself never travels.

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]

Modules used (SURVEY.md 8c "importable here"):
  utils.sh_utils        eval_sh, RGB2SH, SH2RGB                       -> sh_eval.npz
  utils.general_utils   build_rotation, build_scaling_rotation,
                        strip_symmetric  (torch.zeros/device shim)     -> cov3d.npz
  utils.graphics_utils  getWorld2View2, getProjectionMatrix,
                        focal2fov, fov2focal, geom_transform_points    -> camera.npz, geom_transform.npz
  utils.loss_utils      l1_loss, ssim ; utils.image_utils psnr         -> loss.npz
The rasterizer itself (diff_gaussian_rasterization) is absent from the reference, so there
is no fixture for the boundary: "parity unpinned" (see oracle/gsr_ref.c).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np
import torch


def scene_io_fixtures(a):
    """tests/golden/scene_io.npz + scene_io_{cameras,images,points3D}.txt: what the reference's own COLMAP text readers,
    readColmapCameras, getNerfppNorm, loadCam, camera_to_JSON and GaussianModel.capture()/restore() return on inputs written
    here.  plyfile is not installed (an empty stand-in module lets scene/dataset_readers.py import; its PLY
    functions are not called), device="cuda" literals / .cuda() calls are shimmed to stay on the CPU."""
    import importlib.util, struct, tempfile, types, json
    from PIL import Image
    rng = np.random.default_rng(2024)
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    sys.path.insert(1, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))     # simple_knn._C of this repo (import only)
    _zeros, _cuda, _empty_cache = torch.zeros, torch.Tensor.cuda, torch.cuda.empty_cache

    def zeros_cpu(*args, **kw):
        kw.pop("device", None)
        return _zeros(*args, **kw)
    torch.zeros = zeros_cpu
    torch.Tensor.cuda = lambda self, *a_, **k_: self
    torch.cuda.empty_cache = lambda: None
    out = {}
    try:
        spec = importlib.util.spec_from_file_location("ref_colmap_loader", os.path.join(a.ref, "scene", "colmap_loader.py"))
        cl = importlib.util.module_from_spec(spec); spec.loader.exec_module(cl)
        # ---- text model written from the shipped table_ds binaries (+ synthetic poses: images.bin is not in the snapshot) ----
        model_dir = os.path.join(a.ref, "table_ds", "sparse", "0")
        cam = next(iter(cl.read_intrinsics_binary(os.path.join(model_dir, "cameras.bin")).values()))
        f = float(cam.params[0])
        with open(os.path.join(a.out, "scene_io_cameras.txt"), "w") as fh:
            fh.write("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: 2\n")
            fh.write(f"1 PINHOLE {cam.width} {cam.height} {f!r} {f * 1.01!r} {float(cam.params[1])!r} {float(cam.params[2])!r}\n")
            fh.write(f"2 PINHOLE 1600 900 1210.5 1209.25 800.0 450.0\n")
        raw = open(os.path.join(model_dir, "points3D.bin"), "rb").read()
        off, K = 8, 16
        with open(os.path.join(a.out, "scene_io_points3D.txt"), "w") as fh:
            fh.write("# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
            for _ in range(K):
                pid, x, y, z, r, g, b, e = struct.unpack_from("<QdddBBBd", raw, off)
                (track,) = struct.unpack_from("<Q", raw, off + 43)
                tr = struct.unpack_from("<" + "ii" * track, raw, off + 51)
                off += 51 + 8 * track
                fh.write(f"{pid} {x!r} {y!r} {z!r} {r} {g} {b} {e!r} " + " ".join(str(v) for v in tr) + "\n")
        names = ["IMG_0003.JPG", "IMG_0001.JPG", "IMG_0002.JPG", "IMG_0010.JPG", "IMG_0007.JPG"]
        with open(os.path.join(a.out, "scene_io_images.txt"), "w") as fh:
            fh.write("# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
            for i, nm in enumerate(names):
                q = rng.normal(size=4); q /= np.linalg.norm(q)
                t = rng.normal(size=3) * 2.0
                fh.write(f"{i + 1} " + " ".join(repr(float(v)) for v in (*q, *t)) + f" {1 if i != 3 else 2} {nm}\n")
                npts = [3, 0, 5, 1, 2][i]
                fh.write(" ".join(f"{float(rng.uniform(0, 4000))!r} {float(rng.uniform(0, 2200))!r} {int(rng.integers(-1, 500))}" for _ in range(npts)) + "\n")
        ci = cl.read_intrinsics_text(os.path.join(a.out, "scene_io_cameras.txt"))
        ce = cl.read_extrinsics_text(os.path.join(a.out, "scene_io_images.txt"))
        xyz, rgb, err = cl.read_points3D_text(os.path.join(a.out, "scene_io_points3D.txt"))
        out.update(txt_cam_ids=np.array(sorted(ci)), txt_cam_wh=np.array([[ci[k].width, ci[k].height] for k in sorted(ci)]),
                   txt_cam_params=np.array([ci[k].params for k in sorted(ci)]), txt_img_ids=np.array(sorted(ce)),
                   txt_img_q=np.array([ce[k].qvec for k in sorted(ce)]), txt_img_t=np.array([ce[k].tvec for k in sorted(ce)]),
                   txt_img_cam=np.array([ce[k].camera_id for k in sorted(ce)]), txt_img_names=np.array([ce[k].name for k in sorted(ce)]),
                   txt_img_npts=np.array([len(ce[k].point3D_ids) for k in sorted(ce)]),
                   txt_img_xys2=ce[3].xys, txt_img_p3d2=ce[3].point3D_ids, txt_xyz=xyz, txt_rgb=rgb, txt_err=err)
        # ---- camera records, NeRF++ normalisation, training cameras, cameras.json ----
        from scene import dataset_readers as dr
        from utils import camera_utils as cu
        with tempfile.TemporaryDirectory() as td:
            for nm in names:                                  # the images only have to exist and have a size
                w, h = (cam.width, cam.height) if nm != "IMG_0010.JPG" else (1600, 900)
                Image.new("RGB", (w // 8, h // 8), (40, 90, 160)).resize((w, h)).save(os.path.join(td, nm), quality=30)
            infos = dr.readColmapCameras(cam_extrinsics=ce, cam_intrinsics=ci, images_folder=td)
            infos = sorted(infos.copy(), key=lambda x: x.image_name)
            norm = dr.getNerfppNorm(infos)
            out.update(info_names=np.array([c.image_name for c in infos]), info_uid=np.array([c.uid for c in infos]),
                       info_R=np.array([c.R for c in infos]), info_T=np.array([c.T for c in infos]),
                       info_fovx=np.array([c.FovX for c in infos]), info_fovy=np.array([c.FovY for c in infos]),
                       info_wh=np.array([[c.width, c.height] for c in infos]), norm_translate=norm["translate"], norm_radius=norm["radius"])
            res_cases = [(1, 1.0), (2, 1.0), (4, 1.0), (8, 1.0), (-1, 1.0), (-1, 2.0), (800, 1.0), (1000, 1.5), (3, 1.0), (2, 2.0)]
            wh = []
            for res, rs in res_cases:
                args_ = types.SimpleNamespace(resolution=res, data_device="cpu")
                row = []
                for j in (0, 4):                              # a 4032 x 2268 image and the 1600 x 900 one (IMG_0010)
                    c = cu.loadCam(args_, j, infos[j], rs)
                    row.append([c.image_width, c.image_height])
                wh.append(row)
            out.update(res_cases=np.array(res_cases), res_wh=np.array(wh))
            args_ = types.SimpleNamespace(resolution=-1, data_device="cpu")
            cams = cu.cameraList_from_camInfos(infos, 1.0, args_)
            out.update(cam_wvt=np.array([c.world_view_transform.numpy() for c in cams]), cam_full=np.array([c.full_proj_transform.numpy() for c in cams]),
                       cam_center=np.array([c.camera_center.numpy() for c in cams]), cam_wh=np.array([[c.image_width, c.image_height] for c in cams]))
            js = [cu.camera_to_JSON(i, c_) for i, c_ in enumerate(infos)]        # Scene.__init__ passes the CameraInfo records
            out["cameras_json"] = np.array(json.dumps(js))
        # ---- checkpoint tuple: GaussianModel.capture() / restore() ----
        from scene.gaussian_model import GaussianModel
        from arguments import OptimizationParams
        import argparse as _ap
        opt = OptimizationParams(_ap.ArgumentParser())
        g = torch.Generator().manual_seed(4321)
        P, deg = 50, 2
        M = (deg + 1) ** 2
        init = dict(xyz=torch.randn(P, 3, generator=g), f_dc=torch.randn(P, 1, 3, generator=g) * 0.5,
                    f_rest=torch.randn(P, M - 1, 3, generator=g) * 0.1, opacity=torch.randn(P, 1, generator=g) * 2.0,
                    scaling=torch.randn(P, 3, generator=g) * 0.8 - 2.5, rotation=torch.randn(P, 4, generator=g))
        gm = GaussianModel(deg)
        gm.active_sh_degree = 1
        gm._xyz, gm._features_dc, gm._features_rest = (torch.nn.Parameter(init[k].clone()) for k in ("xyz", "f_dc", "f_rest"))
        gm._opacity, gm._scaling, gm._rotation = (torch.nn.Parameter(init[k].clone()) for k in ("opacity", "scaling", "rotation"))
        gm.max_radii2D = torch.rand(P, generator=g) * 30
        gm.spatial_lr_scale = 3.5
        gm.training_setup(opt)
        names_ = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        grads = {k: torch.randn(init[k].shape, generator=g) * 0.01 for k in names_}
        gm.update_learning_rate(7)
        for grp in gm.optimizer.param_groups:
            grp["params"][0].grad = grads[grp["name"]].clone()
        gm.optimizer.step(); gm.optimizer.zero_grad(set_to_none=True)
        gm.xyz_gradient_accum = torch.rand(P, 1, generator=g); gm.denom = (torch.rand(P, 1, generator=g) * 5).floor()
        cap = gm.capture()
        out.update({f"ck_init_{k}": v.numpy() for k, v in init.items()})
        out.update({f"ck_grad_{k}": v.numpy() for k, v in grads.items()})
        out["ck_len"] = len(cap); out["ck_active_sh_degree"] = cap[0]; out["ck_spatial_lr_scale"] = cap[11]
        for idx, k in zip(range(1, 7), ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")):
            out[f"ck_{idx}_{k}"] = cap[idx].detach().numpy().copy()      # capture() hands out the live tensors
        out["ck_7_max_radii2D"] = cap[7].numpy().copy(); out["ck_8_xyz_gradient_accum"] = cap[8].numpy().copy(); out["ck_9_denom"] = cap[9].numpy().copy()
        sd = cap[10]
        out["ck_opt_groups"] = np.array(json.dumps([{k: v for k, v in gp.items()} for gp in sd["param_groups"]]))
        for pid, st in sd["state"].items():
            out[f"ck_opt_state{pid}_step"] = np.array(float(st["step"]))
            out[f"ck_opt_state{pid}_exp_avg"] = st["exp_avg"].numpy().copy(); out[f"ck_opt_state{pid}_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
        # restore() into a fresh model, one more identical step on both: the states must stay equal (pins restore's order of operations)
        import copy
        gm2 = GaussianModel(deg)
        gm2.restore(copy.deepcopy(cap), opt)                  # as after torch.save / torch.load: restore() itself shares the tensors
        for m_ in (gm, gm2):
            m_.update_learning_rate(8)
            for grp in m_.optimizer.param_groups:
                grp["params"][0].grad = grads[grp["name"]].clone() * 0.5
            m_.optimizer.step()
        assert all(torch.equal(a_, b_) for a_, b_ in zip((gm._xyz, gm._opacity, gm._rotation), (gm2._xyz, gm2._opacity, gm2._rotation)))
        out.update({f"ck_after_{k}": getattr(gm2, a_).detach().numpy() for k, a_ in
                    zip(names_, ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"))})
        out["ck_after_lr_xyz"] = np.array([grp["lr"] for grp in gm2.optimizer.param_groups if grp["name"] == "xyz"][0])
    finally:
        torch.zeros, torch.Tensor.cuda, torch.cuda.empty_cache = _zeros, _cuda, _empty_cache
    np.savez_compressed(os.path.join(a.out, "scene_io.npz"), **out)
    print("wrote scene_io.npz,", sorted(n for n in os.listdir(a.out) if n.startswith("scene_io")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="'scene_io': regenerate only the scene_io fixtures")
    a = ap.parse_args()
    sys.path.insert(0, a.ref)
    os.makedirs(a.out, exist_ok=True)
    if a.only == "scene_io":
        scene_io_fixtures(a)
        return
    torch.manual_seed(0)
    rng = np.random.default_rng(0)

    # ---- SH ----
    from utils import sh_utils
    dirs = rng.normal(size=(64, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    sh = rng.normal(0, 0.4, size=(64, 3, 16))          # reference layout for eval_sh: [..., C, coeffs]
    out = {}
    for deg in range(4):
        r = sh_utils.eval_sh(deg, torch.tensor(sh), torch.tensor(dirs)).numpy()
        out[f"rgb_raw_deg{deg}"] = r
        out[f"rgb_clamped_deg{deg}"] = torch.clamp_min(torch.tensor(r) + 0.5, 0.0).numpy()   # gaussian_renderer/__init__.py:78
    rgb = rng.uniform(0, 1, size=(16, 3))
    np.savez(os.path.join(a.out, "sh_eval.npz"), dirs=dirs, sh_view=sh, rgb=rgb,
             rgb2sh=sh_utils.RGB2SH(rgb), sh2rgb=sh_utils.SH2RGB(rgb), C0=sh_utils.C0, C1=sh_utils.C1,
             C2=np.array(sh_utils.C2), C3=np.array(sh_utils.C3), **out)

    # ---- cov3D (device="cuda" is hard-wired in the reference: shim torch.zeros) ----
    from utils import general_utils
    _zeros = torch.zeros

    def zeros_cpu(*args, **kw):
        kw.pop("device", None)
        return _zeros(*args, **kw)
    torch.zeros = zeros_cpu
    try:
        s = np.exp(rng.normal(-2.0, 0.7, size=(64, 3))).astype(np.float32)
        q = rng.normal(size=(64, 4)).astype(np.float32)            # un-normalised on purpose
        qn = q / np.linalg.norm(q, axis=1, keepdims=True)
        res = {}
        for mod in (1.0, 0.5):
            L = general_utils.build_scaling_rotation(torch.tensor(mod * s), torch.tensor(q))
            cov = L @ L.transpose(1, 2)
            res[f"cov6_mod{mod}"] = general_utils.strip_symmetric(cov).numpy()
        Rm = general_utils.build_rotation(torch.tensor(q)).numpy()
    finally:
        torch.zeros = _zeros
    np.savez(os.path.join(a.out, "cov3d.npz"), scales=s, quats_raw=q, quats_unit=qn.astype(np.float32), rot=Rm, **res)

    # ---- camera matrices ----
    from utils import graphics_utils as gu
    cams = {}
    intr = {"table": (3049.779011853469, 4032, 2268), "tiramisu": (3287.4641158882314, 4032, 2268)}
    for name, (f, w, h) in intr.items():
        fx, fy = gu.focal2fov(f, w), gu.focal2fov(f, h)
        cams[f"{name}_fovx"] = fx; cams[f"{name}_fovy"] = fy
        cams[f"{name}_focal_back"] = gu.fov2focal(fx, w)
        cams[f"{name}_proj"] = gu.getProjectionMatrix(0.01, 100.0, fx, fy).numpy()
    poses_R, poses_T, w2v, wvt, full, center = [], [], [], [], [], []
    fovx, fovy = cams["table_fovx"], cams["table_fovy"]
    for i in range(4):
        A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0: Q[:, 0] = -Q[:, 0]
        T = rng.normal(size=3) * 2.0
        trans = np.array([0.1 * i, -0.2, 0.3]); scale = 1.0 + 0.25 * i
        m = gu.getWorld2View2(Q, T, trans, scale)
        # scene/cameras.py:54-57 on CPU
        t_wvt = torch.tensor(m).transpose(0, 1)
        t_proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        t_full = (t_wvt.unsqueeze(0).bmm(t_proj.unsqueeze(0))).squeeze(0)
        t_center = t_wvt.inverse()[3, :3]
        poses_R.append(Q); poses_T.append(T); w2v.append(m); wvt.append(t_wvt.numpy()); full.append(t_full.numpy()); center.append(t_center.numpy())
    cams.update(R=np.array(poses_R), T=np.array(poses_T), translate=np.array([[0.1 * i, -0.2, 0.3] for i in range(4)]),
                scale=np.array([1.0 + 0.25 * i for i in range(4)]), w2v=np.array(w2v), world_view_transform=np.array(wvt),
                full_proj_transform=np.array(full), camera_center=np.array(center),
                w2v_identity=gu.getWorld2View2(np.eye(3), np.array([0.0, 0.0, 3.0])))
    np.savez(os.path.join(a.out, "camera.npz"), **cams)

    # ---- geom_transform_points (the 1e-7 perspective epsilon) ----
    # points placed in front of pose 0 (camera space x,y in +-1, z in 2..6, some behind), moved to world space
    pc = np.stack([rng.uniform(-1, 1, 64), rng.uniform(-0.6, 0.6, 64), rng.uniform(2, 6, 64), np.ones(64)], 1)
    pc[:8, 2] = rng.uniform(-3, 0.15, 8)
    pts = (pc @ np.linalg.inv(wvt[0].astype(np.float64)))[:, :3].astype(np.float32)
    gt = gu.geom_transform_points(torch.tensor(pts), torch.tensor(full[0])).numpy()
    np.savez(os.path.join(a.out, "geom_transform.npz"), points=pts, matrix=full[0], out=gt)

    # ---- loss ----
    from utils import loss_utils, image_utils
    img1 = torch.tensor(rng.uniform(0, 1, size=(3, 64, 64)).astype(np.float32), requires_grad=True)
    img2 = torch.tensor(rng.uniform(0, 1, size=(3, 64, 64)).astype(np.float32))
    l1 = loss_utils.l1_loss(img1, img2)
    ss = loss_utils.ssim(img1, img2)
    loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - ss)                       # train.py:91-92, lambda_dssim arguments/__init__.py:83
    loss.backward()
    ps = image_utils.psnr(img1.detach()[None], img2[None])
    np.savez(os.path.join(a.out, "loss.npz"), img1=img1.detach().numpy(), img2=img2.numpy(), l1=l1.item(), ssim=ss.item(),
             loss=loss.item(), grad=img1.grad.numpy(), psnr=ps.numpy())
    # ---- on-disk formats: bytes cut from the shipped table_ds model + what the reference's loader returns ----
    import importlib.util, struct, shutil
    spec = importlib.util.spec_from_file_location("ref_colmap_loader", os.path.join(a.ref, "scene", "colmap_loader.py"))
    cl = importlib.util.module_from_spec(spec); spec.loader.exec_module(cl)
    model_dir = os.path.join(a.ref, "table_ds", "sparse", "0")
    cams = cl.read_intrinsics_binary(os.path.join(model_dir, "cameras.bin"))
    cam = cams[sorted(cams)[0]]
    cams2 = cl.read_intrinsics_binary(os.path.join(a.ref, "tiramisu_ds", "sparse", "0", "cameras.bin"))
    cam2 = cams2[sorted(cams2)[0]]
    xyz, rgb, err = cl.read_points3D_binary(os.path.join(model_dir, "points3D.bin"))
    shutil.copyfile(os.path.join(model_dir, "cameras.bin"), os.path.join(a.out, "io_cameras.bin"))
    K = 64
    raw = open(os.path.join(model_dir, "points3D.bin"), "rb").read()
    off = 8
    for _ in range(K):                      # walk K variable-length records
        (track,) = struct.unpack_from("<Q", raw, off + 43)
        off += 43 + 8 + 8 * track
    open(os.path.join(a.out, "io_points3D_first64.bin"), "wb").write(struct.pack("<Q", K) + raw[8:off])
    ply = open(os.path.join(model_dir, "points3D.ply"), "rb").read()
    hend = ply.index(b"end_header\n") + len(b"end_header\n")
    header = ply[:hend].replace(b"element vertex %d" % xyz.shape[0], b"element vertex %d" % K)
    open(os.path.join(a.out, "io_points3D_first64.ply"), "wb").write(header + ply[hend:hend + K * 27])
    np.savez(os.path.join(a.out, "io_colmap.npz"), cam_id=cam.id, cam_model=cam.model, cam_width=cam.width, cam_height=cam.height,
             cam_params=cam.params, cam2_params=cam2.params, cam2_width=cam2.width, n_points=xyz.shape[0],
             xyz_first=xyz[:K], rgb_first=rgb[:K], err_first=err[:K], bbox_min=xyz.min(0), bbox_max=xyz.max(0))
    # ---- density control: the reference's GaussianModel on CPU (scene/gaussian_model.py:149-167, 210-213, 258-407) ----
    # plyfile is not installed (only needed by its PLY I/O): an empty stand-in module lets the file import; its
    # simple_knn._C import resolves to this repo's module and is never called here; device="cuda" literals are shimmed.
    import types
    sys.modules.setdefault("plyfile", types.SimpleNamespace(PlyData=None, PlyElement=None))
    sys.path.insert(1, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    _zeros2 = torch.zeros

    def zeros_cpu2(*args, **kw):
        kw.pop("device", None)
        return _zeros2(*args, **kw)
    torch.zeros = zeros_cpu2
    _empty_cache = torch.cuda.empty_cache
    torch.cuda.empty_cache = lambda: None
    try:
        from scene.gaussian_model import GaussianModel
        from arguments import OptimizationParams
        import argparse as _ap
        opt = OptimizationParams(_ap.ArgumentParser())
        g = torch.Generator().manual_seed(1234)
        P, deg = 300, 2
        M = (deg + 1) ** 2
        init = dict(xyz=torch.randn(P, 3, generator=g), f_dc=torch.randn(P, 1, 3, generator=g) * 0.5,
                    f_rest=torch.randn(P, M - 1, 3, generator=g) * 0.1, opacity=torch.randn(P, 1, generator=g) * 2.0,
                    scaling=torch.randn(P, 3, generator=g) * 0.8 - 2.5, rotation=torch.randn(P, 4, generator=g))
        gm = GaussianModel(deg)
        gm._xyz, gm._features_dc, gm._features_rest = (torch.nn.Parameter(init[k].clone()) for k in ("xyz", "f_dc", "f_rest"))
        gm._opacity, gm._scaling, gm._rotation = (torch.nn.Parameter(init[k].clone()) for k in ("opacity", "scaling", "rotation"))
        gm.max_radii2D = torch.zeros(P)
        gm.spatial_lr_scale = 2.5
        gm.training_setup(opt)
        names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        out = {f"init_{k}": v.numpy() for k, v in init.items()}
        out["lr_at"] = np.array([gm.xyz_scheduler_args(i) for i in (0, 1, 100, 7000, 30000, 40000)])
        # two Adam steps with fixed gradients so that the moments are non-trivial
        grads = [{k: torch.randn(init[k].shape, generator=g) * 0.01 for k in names} for _ in range(2)]
        for step, gr in enumerate(grads):
            gm.update_learning_rate(step + 1)
            for grp in gm.optimizer.param_groups:
                grp["params"][0].grad = gr[grp["name"]].clone()
            gm.optimizer.step(); gm.optimizer.zero_grad(set_to_none=True)
            for k in names:
                out[f"grad{step}_{k}"] = gr[k].numpy()
        # densification statistics from three "views"
        stats = []
        for v in range(3):
            vs = torch.zeros(P, 3); vs.grad = torch.randn(P, 3, generator=g) * 0.0005
            vis = torch.rand(P, generator=g) < 0.7
            radii = (torch.rand(P, generator=g) * 40).floor()
            gm.max_radii2D[vis] = torch.max(gm.max_radii2D[vis], radii[vis])
            gm.add_densification_stats(vs, vis)
            stats.append((vs.grad.numpy(), vis.numpy(), radii.numpy()))
        for v, (a_, b_, c_) in enumerate(stats):
            out[f"view{v}_grad"], out[f"view{v}_vis"], out[f"view{v}_radii"] = a_, b_, c_
        torch.manual_seed(77)                      # the split samples come from the global generator in the reference
        gm.densify_and_prune(0.0002, 0.005, 4.0, 20)
        state = lambda tag: {**{f"{tag}_{k}": getattr(gm, a_).detach().numpy() for k, a_ in
                               zip(names, ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"))},
                             **{f"{tag}_m_{grp['name']}": gm.optimizer.state[grp["params"][0]]["exp_avg"].numpy() for grp in gm.optimizer.param_groups},
                             **{f"{tag}_v_{grp['name']}": gm.optimizer.state[grp["params"][0]]["exp_avg_sq"].numpy() for grp in gm.optimizer.param_groups}}
        out.update(state("dens"))
        gm.reset_opacity()
        out.update(state("reset"))
        np.savez_compressed(os.path.join(a.out, "densify.npz"), **out)
    finally:
        torch.zeros = _zeros2
        torch.cuda.empty_cache = _empty_cache

    # the whole SfM cloud of table_ds (17 618 points, a data file of the reference's dataset): a realistic depth /
    # footprint distribution for the GPU parity tests (BASELINE config 2 names this scene)
    shutil.copyfile(os.path.join(model_dir, "points3D.ply"), os.path.join(a.out, "table_points3D.ply"))
    # the SfM cloud of tiramisu_ds (33 730 points): BASELINE config 4 names this scene
    shutil.copyfile(os.path.join(a.ref, "tiramisu_ds", "sparse", "0", "points3D.ply"), os.path.join(a.out, "tiramisu_points3D.ply"))
    scene_io_fixtures(a)
    print("wrote", sorted(os.listdir(a.out)))


if __name__ == "__main__":
    main()
