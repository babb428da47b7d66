"""rvpt_amd — MI355X (gfx950) path-trace backend for RVPT's compute pass.

The product is the C-ABI shared library (include/rvpt_hip.h, rvpt_amd/csrc/); this package is the thin
Python host layer above it: ctypes bindings (native), a mirror of the reference's `class RVPT`
upload/update/draw interface (renderer), host scene + camera PODs (scene, camera), and the
one-process-per-GPU tile partition with an RCCL gather (distributed).
"""
from . import camera, scene  # noqa: F401
from .camera import Camera  # noqa: F401
from .native import NativeError, lib_path, load  # noqa: F401
from .renderer import RVPT, RenderSettings  # noqa: F401

__all__ = ["RVPT", "RenderSettings", "Camera", "scene", "camera", "NativeError", "load", "lib_path"]
