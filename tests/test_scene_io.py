"""CPU: scene_io.py against tests/golden/scene_io.npz -- what the reference's own loaders / GaussianModel return for the
same inputs (oracle/make_golden.py scene_io_fixtures): COLMAP text readers, camera records, NeRF++ normalisation, the
training-resolution rule of loadCam, camera matrices of the loaded cameras, cameras.json entries, checkpoint tuple."""
import json
import os

import numpy as np
import pytest
import torch

from gaussian_transformer_amd import io as gio
from gaussian_transformer_amd import scene_io as sio
from gaussian_transformer_amd.densify import DensityController, OptimizationParams
from gaussian_transformer_amd.model import GaussianParams

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "scene_io.npz"), allow_pickle=False)


def test_colmap_text_readers(gold):
    cams = sio.read_cameras_text(os.path.join(G, "scene_io_cameras.txt"))
    assert sorted(cams) == gold["txt_cam_ids"].tolist()
    np.testing.assert_array_equal([[cams[k].width, cams[k].height] for k in sorted(cams)], gold["txt_cam_wh"])
    np.testing.assert_array_equal([cams[k].params for k in sorted(cams)], gold["txt_cam_params"])
    assert all(c.model == "PINHOLE" for c in cams.values())
    imgs = sio.read_images_text(os.path.join(G, "scene_io_images.txt"))
    ids = sorted(imgs)
    assert ids == gold["txt_img_ids"].tolist()
    np.testing.assert_array_equal([imgs[k].qvec for k in ids], gold["txt_img_q"])
    np.testing.assert_array_equal([imgs[k].tvec for k in ids], gold["txt_img_t"])
    assert [imgs[k].camera_id for k in ids] == gold["txt_img_cam"].tolist()
    assert [imgs[k].name for k in ids] == gold["txt_img_names"].tolist()
    assert [len(imgs[k].point3D_ids) for k in ids] == gold["txt_img_npts"].tolist()       # includes an image without 2D points
    np.testing.assert_array_equal(imgs[3].xys, gold["txt_img_xys2"]); np.testing.assert_array_equal(imgs[3].point3D_ids, gold["txt_img_p3d2"])
    xyz, rgb, err = sio.read_points3D_text(os.path.join(G, "scene_io_points3D.txt"))
    np.testing.assert_array_equal(xyz, gold["txt_xyz"]); np.testing.assert_array_equal(rgb, gold["txt_rgb"]); np.testing.assert_array_equal(err, gold["txt_err"])
    # the text points are the first 16 of the shipped binary model: both readers must agree
    bx, br, be = gio.read_points3D_binary(os.path.join(G, "io_points3D_first64.bin"))
    np.testing.assert_array_equal(xyz, bx[:16]); np.testing.assert_array_equal(rgb, br[:16]); np.testing.assert_array_equal(err, be[:16])


def test_text_reader_rejects_what_the_reference_rejects(tmp_path):
    p = tmp_path / "cameras.txt"
    p.write_text("1 SIMPLE_PINHOLE 4032 2268 3049.7 2016 1134\n")
    with pytest.raises(AssertionError, match="PINHOLE"):
        sio.read_cameras_text(str(p))


def _infos(gold):
    cams = sio.read_cameras_text(os.path.join(G, "scene_io_cameras.txt"))
    imgs = sio.read_images_text(os.path.join(G, "scene_io_images.txt"))
    infos = sio.colmap_camera_infos(imgs, cams, "/nonexistent/images")
    return sorted(infos, key=lambda c: c.image_name)


def test_camera_records_and_nerfpp_norm(gold):
    infos = _infos(gold)
    assert [c.image_name for c in infos] == gold["info_names"].tolist() and [c.uid for c in infos] == gold["info_uid"].tolist()
    np.testing.assert_array_equal([c.R for c in infos], gold["info_R"]); np.testing.assert_array_equal([c.T for c in infos], gold["info_T"])
    np.testing.assert_array_equal([c.FovX for c in infos], gold["info_fovx"]); np.testing.assert_array_equal([c.FovY for c in infos], gold["info_fovy"])
    np.testing.assert_array_equal([[c.width, c.height] for c in infos], gold["info_wh"])
    n = sio.nerfpp_norm(infos)
    np.testing.assert_allclose(n["translate"], gold["norm_translate"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(n["radius"], gold["norm_radius"], rtol=1e-6)


def test_training_resolution_rule(gold):
    sizes = ((4032, 2268), (1600, 900))
    for (res, rs), row in zip(gold["res_cases"], gold["res_wh"]):
        res = int(res) if float(res).is_integer() else float(res)
        for (w, h), want in zip(sizes, row):
            assert list(sio.training_resolution(w, h, res, float(rs))) == want.tolist(), (res, rs, w, h)


def test_loaded_camera_matrices_and_json(gold):
    infos = _infos(gold)
    for j, c in enumerate(infos):
        cam, (w, h) = sio.load_camera(c, j, resolution=-1, resolution_scale=1.0)
        assert [w, h] == gold["cam_wh"][j].tolist() == [cam.image_width, cam.image_height]
        np.testing.assert_allclose(cam.world_view_transform, gold["cam_wvt"][j], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cam.full_proj_transform, gold["cam_full"][j], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center, gold["cam_center"][j], rtol=1e-5, atol=1e-6)
    want = json.loads(str(gold["cameras_json"]))
    got = [sio.camera_to_json(i, c.R, c.T, c.FovX, c.FovY, c.width, c.height, c.image_name) for i, c in enumerate(infos)]
    assert [sorted(d) for d in got] == [sorted(d) for d in want]
    for a, b in zip(got, want):
        for k in a:
            if isinstance(a[k], (list, float)):
                np.testing.assert_allclose(a[k], b[k], rtol=1e-12, atol=1e-12)
            else:
                assert a[k] == b[k], k


def _controller_from_fixture(gold, adam="torch"):
    m = GaussianParams(2)
    m.active_sh_degree = 1
    t = lambda k: torch.tensor(gold[f"ck_init_{k}"])
    m._xyz, m._features_dc, m._features_rest = t("xyz"), t("f_dc"), t("f_rest")
    m._opacity, m._scaling, m._rotation = t("opacity"), t("scaling"), t("rotation")
    ctl = DensityController(m, OptimizationParams(), spatial_lr_scale=3.5)
    m.max_radii2D = torch.tensor(gold["ck_7_max_radii2D"])
    ctl.update_learning_rate(7)
    for g in ctl.optimizer.param_groups:
        g["params"][0].grad = torch.tensor(gold[f"ck_grad_{g['name']}"])
    ctl.optimizer.step(); ctl.optimizer.zero_grad(set_to_none=True)
    m.xyz_gradient_accum = torch.tensor(gold["ck_8_xyz_gradient_accum"]); m.denom = torch.tensor(gold["ck_9_denom"])
    return ctl


def _check_capture(cap, gold):
    assert len(cap) == int(gold["ck_len"]) == len(sio.CAPTURE_FIELDS) == 12
    assert cap[0] == int(gold["ck_active_sh_degree"]) and cap[11] == float(gold["ck_spatial_lr_scale"])
    for idx, k in zip(range(1, 7), ("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity")):     # the tuple's own order
        np.testing.assert_array_equal(cap[idx].detach().numpy(), gold[f"ck_{idx}_{k}"])
    np.testing.assert_array_equal(cap[7].numpy(), gold["ck_7_max_radii2D"])
    np.testing.assert_array_equal(cap[8].numpy(), gold["ck_8_xyz_gradient_accum"]); np.testing.assert_array_equal(cap[9].numpy(), gold["ck_9_denom"])
    sd = cap[10]
    want_groups = json.loads(str(gold["ck_opt_groups"]))
    assert [g["name"] for g in sd["param_groups"]] == [g["name"] for g in want_groups]
    for g, w in zip(sd["param_groups"], want_groups):
        assert g["params"] == w["params"] and g["eps"] == w["eps"] and tuple(g["betas"]) == tuple(w["betas"])
        assert g["lr"] == pytest.approx(w["lr"], rel=1e-12)
    for pid in range(6):
        st = sd["state"][pid]
        assert float(st["step"]) == float(gold[f"ck_opt_state{pid}_step"])
        np.testing.assert_array_equal(st["exp_avg"].numpy(), gold[f"ck_opt_state{pid}_exp_avg"])
        np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), gold[f"ck_opt_state{pid}_exp_avg_sq"])


def test_checkpoint_capture_restore_match_the_reference(gold, tmp_path):
    ctl = _controller_from_fixture(gold)
    cap = sio.capture(ctl)
    _check_capture(cap, gold)                               # same tuple as GaussianModel.capture() after the same history
    # through a checkpoint file (train.py:130-132 / :37-39), loaded without unpickling code
    path = str(tmp_path / "chkpnt7.pth")
    sio.save_checkpoint(path, ctl, 7)
    ctl2, first_iter = sio.load_checkpoint(path, OptimizationParams())
    assert first_iter == 7
    _check_capture(sio.capture(ctl2), gold)
    # one more step after restore: equals what the reference's restored model does
    ctl2.update_learning_rate(8)
    for g in ctl2.optimizer.param_groups:
        g["params"][0].grad = torch.tensor(gold[f"ck_grad_{g['name']}"]) * 0.5
    ctl2.optimizer.step()
    m = ctl2.model
    for k, attr in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"),
                       ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")):
        np.testing.assert_array_equal(getattr(m, attr).detach().numpy(), gold[f"ck_after_{k}"])
    assert [g["lr"] for g in ctl2.optimizer.param_groups if g["name"] == "xyz"][0] == pytest.approx(float(gold["ck_after_lr_xyz"]), rel=1e-12)


def test_read_colmap_scene_from_text_model(tmp_path, gold):
    """readColmapSceneInfo's fall-backs: no .bin files -> the text pair; points3D.ply written from points3D.txt; llffhold split."""
    sp = tmp_path / "sparse" / "0"
    sp.mkdir(parents=True)
    for n in ("cameras", "images", "points3D"):
        (sp / f"{n}.txt").write_bytes(open(os.path.join(G, f"scene_io_{n}.txt"), "rb").read())
    sc = sio.read_colmap_scene(str(tmp_path), eval=True, llffhold=2)
    names = gold["info_names"].tolist()
    assert [c.image_name for c in sc.test_cameras] == names[0::2] and [c.image_name for c in sc.train_cameras] == names[1::2]
    assert os.path.exists(sc.ply_path) and sc.point_cloud.points.shape == (16, 3)
    np.testing.assert_allclose(sc.point_cloud.points, gold["txt_xyz"].astype(np.float32))
    np.testing.assert_allclose(sc.point_cloud.colors, gold["txt_rgb"] / 255.0)
    assert sc.nerf_normalization["radius"] > 0
