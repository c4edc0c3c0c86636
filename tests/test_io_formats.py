"""On-disk formats (gaussian_transformer_amd/io.py) pinned by bytes cut from the reference's shipped table_ds
COLMAP model and by what the reference's own scene/colmap_loader.py returns for them (tests/golden/io_*)."""
import os

import numpy as np
import pytest

from gaussian_transformer_amd import io as gio

G = os.path.join(os.path.dirname(__file__), "golden")


def test_colmap_binary_readers_match_reference_loader():
    d = np.load(os.path.join(G, "io_colmap.npz"))
    cams = gio.read_cameras_binary(os.path.join(G, "io_cameras.bin"))
    cam = cams[int(d["cam_id"])]
    assert cam.model == str(d["cam_model"]) == "SIMPLE_PINHOLE"
    assert (cam.width, cam.height) == (int(d["cam_width"]), int(d["cam_height"])) == (4032, 2268)
    np.testing.assert_array_equal(cam.params, d["cam_params"])
    assert cam.params[0] == pytest.approx(3049.779011853469)          # SURVEY 8c known-answer anchor
    xyz, rgb, err = gio.read_points3D_binary(os.path.join(G, "io_points3D_first64.bin"))
    np.testing.assert_array_equal(xyz, d["xyz_first"])
    np.testing.assert_array_equal(rgb, d["rgb_first"])
    np.testing.assert_array_equal(err, d["err_first"])


def test_point_cloud_ply_reader_agrees_with_colmap_points():
    """points3D.ply was written by the reference from points3D.bin (scene/dataset_readers.py:157-165):
    same cloud as float32 xyz + uint8 rgb."""
    d = np.load(os.path.join(G, "io_colmap.npz"))
    pc = gio.fetch_point_cloud(os.path.join(G, "io_points3D_first64.ply"))
    np.testing.assert_array_equal(pc.points.astype(np.float32), d["xyz_first"].astype(np.float32))
    np.testing.assert_allclose(pc.colors, d["rgb_first"] / 255.0, atol=1e-12)
    assert pc.normals.shape == (64, 3) and not pc.normals.any()


def test_point_cloud_and_gaussian_ply_round_trips(tmp_path):
    rng = np.random.default_rng(0)
    xyz = rng.normal(size=(50, 3)).astype(np.float32); rgb = rng.integers(0, 256, (50, 3))
    p = str(tmp_path / "pc.ply")
    gio.store_point_cloud(p, xyz, rgb)
    pc = gio.fetch_point_cloud(p)
    np.testing.assert_array_equal(pc.points, xyz); np.testing.assert_allclose(pc.colors, rgb / 255.0)
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\n")
    # Gaussian PLY: channel-major f_dc / f_rest on disk, coefficient-major in the model
    P, deg = 20, 2
    M = (deg + 1) ** 2
    par = dict(xyz=rng.normal(size=(P, 3)), features_dc=rng.normal(size=(P, 1, 3)), features_rest=rng.normal(size=(P, M - 1, 3)),
               opacity=rng.normal(size=(P, 1)), scaling=rng.normal(size=(P, 3)), rotation=rng.normal(size=(P, 4)))
    par = {k: v.astype(np.float32) for k, v in par.items()}
    g = str(tmp_path / "point_cloud.ply")
    gio.save_gaussians(g, **par)
    v = gio.read_ply_vertices(g)
    assert list(v.dtype.names) == gio.gaussian_attribute_names(3, 3 * (M - 1))
    np.testing.assert_array_equal(v["f_rest_0"], par["features_rest"][:, 0, 0])          # channel 0, coefficient 1
    np.testing.assert_array_equal(v["f_rest_1"], par["features_rest"][:, 1, 0])          # channel 0, coefficient 2
    np.testing.assert_array_equal(v[f"f_rest_{M - 1}"], par["features_rest"][:, 0, 1])   # channel 1, coefficient 1
    back = gio.load_gaussians(g, deg)
    for k in par:
        np.testing.assert_array_equal(back[k], par[k])
    with pytest.raises(ValueError):
        gio.load_gaussians(g, deg + 1)


def test_images_binary_round_trip(tmp_path):
    import struct
    p = str(tmp_path / "images.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<i", 7) + struct.pack("<dddd", 1, 0, 0, 0) + struct.pack("<ddd", 0.5, -1, 2) + struct.pack("<i", 1))
        f.write(b"img_0007.jpg\x00" + struct.pack("<Q", 2) + struct.pack("<ddq", 1.5, 2.5, 11) + struct.pack("<ddq", 3.5, 4.5, -1))
    im = gio.read_images_binary(p)[7]
    assert im.name == "img_0007.jpg" and im.camera_id == 1 and im.point3D_ids.tolist() == [11, -1]
    np.testing.assert_array_equal(im.xys, [[1.5, 2.5], [3.5, 4.5]])
    np.testing.assert_allclose(gio.qvec2rotmat(im.qvec), np.eye(3))


def test_model_from_point_cloud_and_ply_checkpoint(tmp_path):
    """create_from_pcd (scene/gaussian_model.py:124-146) on the fixture cloud with an injected exact 3-NN (the HIP
    distCUDA2 itself is covered by tests/test_gpu_knn.py), then save_ply -> load_ply."""
    import torch
    from scipy.spatial import cKDTree
    from gaussian_transformer_amd.model import GaussianParams
    pc = gio.fetch_point_cloud(os.path.join(G, "io_points3D_first64.ply"))

    def dist2(p):
        d, _ = cKDTree(p.numpy()).query(p.numpy(), k=4)
        return torch.tensor((d[:, 1:] ** 2).mean(1), dtype=torch.float32)

    m = GaussianParams.create_from_pcd(pc, 3, "cpu", dist2_fn=dist2)
    assert m.active_sh_degree == 0 and m._features_rest.shape == (64, 15, 3) and not m._features_rest.any()
    np.testing.assert_allclose(m._features_dc[:, 0].detach().numpy() * 0.28209479177387814 + 0.5, pc.colors, atol=1e-6)
    np.testing.assert_allclose(torch.exp(m._scaling[:, 0]).detach().numpy() ** 2, dist2(torch.tensor(pc.points, dtype=torch.float32)).numpy(), rtol=1e-5)
    np.testing.assert_allclose(m.get_opacity.detach().numpy(), 0.1, atol=1e-6)
    p = str(tmp_path / "point_cloud.ply")
    m.save_ply(p)
    back = GaussianParams.load_ply(p, 3, "cpu")
    for a, b in zip(m.parameters(), back.parameters()):
        assert torch.equal(a.detach(), b.detach())
