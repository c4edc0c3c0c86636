"""The viewer bridge of the training loop (last item): a non-blocking TCP listener that takes JSON camera
messages from the SIBR remote viewer and answers each with the raw RGB bytes of a render plus a verification string.

Restates the wire protocol (not the code) of the reference's gaussian_renderer/network_gui.py:26-86 and its use in
train.py:52-65:
  viewer -> trainer   4-byte little-endian length, then a UTF-8 JSON object with resolution_x / resolution_y (0 x 0 = no
                      render wanted), train, fov_y, fov_x, z_near, z_far, shs_python, rot_scale_python, keep_alive,
                      scaling_modifier, view_matrix[16], view_projection_matrix[16]
  trainer -> viewer   the image bytes (H x W x 3, uint8; omitted when no render was wanted), then a 4-byte little-endian
                      length and that many ASCII bytes (the dataset's source path)
The viewer's matrices use the OpenGL axes: the y and z columns of the view matrix and the y column of the view-projection
matrix are negated on arrival (:69-73); the camera centre is row 3 of the inverted view matrix (scene/cameras.py:69-70).
Differences from the reference: the state lives in an object instead of module globals (several trainers per process, one
per GPU), nothing here touches a device -- the caller renders -- and a peer that hangs up mid-message raises ConnectionError
instead of yielding a JSON decoding error.  Polled from the training thread, as the reference does: no thread of its own.
"""
from __future__ import annotations

import json
import socket
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class MiniCam:
    """scene/cameras.py:59-70 -- what render() needs from a viewer camera (numpy; render.TorchCamera moves it to the device)."""
    image_width: int
    image_height: int
    FoVy: float
    FoVx: float
    znear: float
    zfar: float
    world_view_transform: np.ndarray     # [4,4] float32, the reference's transposed layout
    full_proj_transform: np.ndarray      # [4,4]
    camera_center: np.ndarray            # [3]


@dataclass
class ViewerRequest:
    camera: Optional[MiniCam]            # None: the viewer asked for no image (resolution 0 x 0)
    do_training: Optional[bool] = None
    convert_SHs_python: Optional[bool] = None
    compute_cov3D_python: Optional[bool] = None
    keep_alive: Optional[bool] = None
    scaling_modifier: Optional[float] = None


def parse_message(message: dict) -> ViewerRequest:
    """The body of receive() (:57-86)."""
    width, height = message["resolution_x"], message["resolution_y"]
    if width == 0 or height == 0:
        return ViewerRequest(None)
    wvt = np.asarray(message["view_matrix"], dtype=np.float32).reshape(4, 4).copy()
    wvt[:, 1] = -wvt[:, 1]
    wvt[:, 2] = -wvt[:, 2]
    full = np.asarray(message["view_projection_matrix"], dtype=np.float32).reshape(4, 4).copy()
    full[:, 1] = -full[:, 1]
    cam = MiniCam(int(width), int(height), message["fov_y"], message["fov_x"], message["z_near"], message["z_far"], wvt, full,
                  np.linalg.inv(wvt)[3, :3].astype(np.float32))
    return ViewerRequest(cam, bool(message["train"]), bool(message["shs_python"]), bool(message["rot_scale_python"]),
                         bool(message["keep_alive"]), message["scaling_modifier"])


def image_to_bytes(image) -> bytes:
    """[3,H,W] float image (torch tensor or array) -> the H x W x 3 uint8 bytes the viewer expects (train.py:60):
    clamp to [0, 1], times 255, truncated."""
    a = image.detach().cpu().numpy() if hasattr(image, "detach") else np.asarray(image)
    return np.ascontiguousarray((np.clip(a, 0.0, 1.0) * 255.0).astype(np.uint8).transpose(1, 2, 0)).tobytes()


class NetworkGUI:
    def __init__(self, host: str = "127.0.0.1", port: int = 6009):
        """init() of the reference: bind, listen, never block on accept."""
        self.listener = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.listener.bind((host, port))
        self.listener.listen()
        self.listener.settimeout(0)
        self.host, self.port = self.listener.getsockname()
        self.conn: Optional[socket.socket] = None
        self.addr = None

    def try_connect(self) -> bool:
        try:
            self.conn, self.addr = self.listener.accept()
            self.conn.settimeout(None)
            return True
        except (BlockingIOError, socket.timeout, OSError):
            return False

    def _recv_exact(self, n: int) -> bytes:
        buf = b""
        while len(buf) < n:
            chunk = self.conn.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("viewer closed the connection")
            buf += chunk
        return buf

    def read(self) -> dict:
        length = int.from_bytes(self._recv_exact(4), "little")
        return json.loads(self._recv_exact(length).decode("utf-8"))

    def receive(self) -> ViewerRequest:
        return parse_message(self.read())

    def send(self, message_bytes: Optional[bytes], verify: str) -> None:
        if message_bytes is not None:
            self.conn.sendall(message_bytes)
        self.conn.sendall(len(verify).to_bytes(4, "little"))
        self.conn.sendall(bytes(verify, "ascii"))

    def close(self) -> None:
        for s in (self.conn, self.listener):
            try:
                if s is not None:
                    s.close()
            except OSError:
                pass
        self.conn = None

    def serve(self, render_fn, verify: str, iteration: int, final_iteration: int) -> Tuple[Optional[dict], bool]:
        """The polling block of train.py:52-65: connect if nobody is connected; while a viewer is connected answer its
        requests, and return to training once it asks for it.  render_fn(request) -> [3,H,W] image.  Returns
        (pipeline overrides of the last request or None, whether a viewer is still connected)."""
        last = None
        if self.conn is None:
            self.try_connect()
        while self.conn is not None:
            try:
                req = self.receive()
                payload = image_to_bytes(render_fn(req)) if req.camera is not None else None
                self.send(payload, verify)
                if req.camera is not None:
                    last = {"convert_SHs_python": req.convert_SHs_python, "compute_cov3D_python": req.compute_cov3D_python}
                if req.do_training and (iteration < final_iteration or not req.keep_alive):
                    break
            except Exception:
                self.conn = None
        return last, self.conn is not None
