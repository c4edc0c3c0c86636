"""CPU: the viewer bridge (network_gui.py) over loopback against a stand-in viewer that speaks the reference's wire protocol
(gaussian_renderer/network_gui.py:43-86, train.py:52-65).  Parity unpinned: the SIBR viewer is an empty submodule; the
message fields and the matrix sign flips are read off the reference's receive()."""
import json
import socket
import threading

import numpy as np

from gaussian_transformer_amd import network_gui as ng
from gaussian_transformer_amd.camera import look_at_camera


def _message(cam, train=True, keep_alive=False, w=None, h=None):
    gl_view = cam.world_view_transform.copy(); gl_view[:, 1] *= -1; gl_view[:, 2] *= -1      # what an OpenGL viewer sends
    gl_full = cam.full_proj_transform.copy(); gl_full[:, 1] *= -1
    return {"resolution_x": cam.image_width if w is None else w, "resolution_y": cam.image_height if h is None else h, "train": train,
            "fov_y": cam.FoVy, "fov_x": cam.FoVx, "z_near": 0.01, "z_far": 100.0, "shs_python": False, "rot_scale_python": True,
            "keep_alive": keep_alive, "scaling_modifier": 0.5, "view_matrix": gl_view.reshape(-1).tolist(),
            "view_projection_matrix": gl_full.reshape(-1).tolist()}


def test_message_parsing_matches_the_reference_conventions():
    cam = look_at_camera((0.3, -0.2, -4.0), (0.0, 0.0, 1.0), (0.0, -1.0, 0.0), 0.9, 64, 48)
    req = ng.parse_message(_message(cam))
    np.testing.assert_array_equal(req.camera.world_view_transform, cam.world_view_transform)        # the flips are undone
    np.testing.assert_array_equal(req.camera.full_proj_transform, cam.full_proj_transform)
    np.testing.assert_allclose(req.camera.camera_center, cam.camera_center, atol=1e-5)
    assert (req.camera.image_width, req.camera.image_height) == (64, 48)
    assert req.do_training is True and req.convert_SHs_python is False and req.compute_cov3D_python is True
    assert req.keep_alive is False and req.scaling_modifier == 0.5
    assert ng.parse_message(_message(cam, w=0, h=0)).camera is None                                  # "no image wanted"


def test_round_trip_with_a_stand_in_viewer():
    gui = ng.NetworkGUI("127.0.0.1", 0)
    cam = look_at_camera((0.0, 0.0, -3.0), (0.0, 0.0, 1.0), (0.0, -1.0, 0.0), 0.8, 16, 12)
    got = {}

    def viewer():
        s = socket.create_connection((gui.host, gui.port))
        for k, msg in enumerate((_message(cam, train=False), _message(cam, w=0, h=0, train=False), _message(cam, train=True))):
            raw = json.dumps(msg).encode("utf-8")
            s.sendall(len(raw).to_bytes(4, "little") + raw)
            n_img = 16 * 12 * 3 if msg["resolution_x"] else 0
            buf = b""
            while len(buf) < n_img + 4:
                buf += s.recv(65536)
            vlen = int.from_bytes(buf[n_img:n_img + 4], "little")
            while len(buf) < n_img + 4 + vlen:
                buf += s.recv(65536)
            got[k] = (buf[:n_img], buf[n_img + 4:].decode("ascii"))
        s.close()

    t = threading.Thread(target=viewer); t.start()
    image = np.linspace(-0.2, 1.2, 3 * 12 * 16, dtype=np.float32).reshape(3, 12, 16)
    calls = []
    import time
    deadline = time.time() + 10
    while gui.conn is None and time.time() < deadline:
        gui.try_connect(); time.sleep(0.01)
    assert gui.conn is not None
    last, connected = gui.serve(lambda req: (calls.append(req), image)[1], "/data/scene", iteration=5, final_iteration=30000)
    t.join(10)
    assert connected and len(calls) == 2                     # the third request has train=True: back to the training loop
    want = (np.clip(image, 0, 1) * 255).astype(np.uint8).transpose(1, 2, 0).tobytes()
    assert got[0] == (want, "/data/scene") and got[1] == (b"", "/data/scene") and got[2] == (want, "/data/scene")
    assert last == {"convert_SHs_python": False, "compute_cov3D_python": True}
    _, connected = gui.serve(lambda req: image, "/data/scene", 6, 30000)     # the viewer hung up: the loop notices and moves on
    assert not connected
    gui.close()
