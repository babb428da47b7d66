"""Host camera: produces the 80-byte block the kernel consumes.

Mirror of the reference Camera (src/rvpt/camera.h:14-57, camera.cpp): translation, Euler rotation in
degrees, fov 90, ortho scale 4, mode 0; get_data() returns the 4 columns of the camera-to-world matrix
followed by (aspect, radians(fov), scale, 0) (camera.cpp:55-66).  The matrix is
T * R(UP, rot.x) * R(RIGHT, rot.y) * R(FORWARD, rot.z)  (construct_camera_matrix, camera.cpp:17-25).
"""
from __future__ import annotations

import math

import numpy as np

RIGHT = (1.0, 0.0, 0.0)
UP = (0.0, 1.0, 0.0)
FORWARD = (0.0, 0.0, 1.0)


def _rotation(angle_rad: float, axis) -> np.ndarray:
    """4x4 rotation about a unit axis (right-handed, same convention as glm::rotate)."""
    x, y, z = axis
    c, s = math.cos(angle_rad), math.sin(angle_rad)
    t = 1.0 - c
    return np.array([
        [c + t * x * x, t * x * y - s * z, t * x * z + s * y, 0.0],
        [t * x * y + s * z, c + t * y * y, t * y * z - s * x, 0.0],
        [t * x * z - s * y, t * y * z + s * x, c + t * z * z, 0.0],
        [0.0, 0.0, 0.0, 1.0],
    ])


def construct_camera_matrix(translation, rotation_deg) -> np.ndarray:
    """camera.cpp:17-25 — returns a 4x4 (row = output component) float64 matrix."""
    m = np.eye(4)
    m[:3, 3] = translation
    m = m @ _rotation(math.radians(rotation_deg[0]), UP)
    m = m @ _rotation(math.radians(rotation_deg[1]), RIGHT)
    m = m @ _rotation(math.radians(rotation_deg[2]), FORWARD)
    return m


class Camera:
    def __init__(self, aspect: float):
        self.aspect = float(aspect)
        self.fov = 90.0  # camera.h:45
        self.scale = 4.0
        self.mode = 0
        self.translation = np.zeros(3)
        self.rotation = np.zeros(3)
        self.vertical_view_angle_clamp = False

    # camera.cpp:29-39
    def translate(self, delta) -> None:
        m = construct_camera_matrix(self.translation, self.rotation)
        self.translation = self.translation + (m @ np.array([*delta, 0.0]))[:3]

    def rotate(self, delta) -> None:
        self.rotation = self.rotation + np.asarray(delta, dtype=np.float64)
        if self.vertical_view_angle_clamp:
            self.rotation[1] = min(90.0, max(-90.0, self.rotation[1]))

    def set_fov(self, fov: float) -> None:
        self.fov = float(fov)

    def set_scale(self, scale: float) -> None:
        self.scale = float(scale)

    def set_camera_mode(self, mode: int) -> None:
        self.mode = int(mode)

    def get_camera_mode(self) -> int:
        return self.mode

    def get_camera_matrix(self) -> np.ndarray:
        return construct_camera_matrix(self.translation, self.rotation)

    def get_data(self) -> np.ndarray:
        """float32[20]: matrix columns 0..3 then (aspect, radians(fov), scale, 0) (camera.cpp:55-66).
        Cached until a camera parameter changes (the per-frame host loop calls this every frame)."""
        key = (self.translation[0], self.translation[1], self.translation[2], self.rotation[0], self.rotation[1],
               self.rotation[2], self.fov, self.scale, self.aspect)
        cached = getattr(self, "_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        out = self._compute_data()
        out.setflags(write=False)
        self._cache = (key, out)
        return out

    def _compute_data(self) -> np.ndarray:
        m = self.get_camera_matrix()
        out = np.zeros(20, dtype=np.float32)
        out[0:16] = m.T.reshape(16)  # column-major
        out[16] = self.aspect
        out[17] = math.radians(self.fov)
        out[18] = self.scale
        return out
