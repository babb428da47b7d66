"""rvpt_amd — MI355X (gfx950) path-trace backend for RVPT's compute pass.

The product is the C-ABI shared library (include/rvpt_hip.h, rvpt_amd/csrc/); this package is the thin
Python host layer above it: ctypes bindings (native), a mirror of the reference's `class RVPT`
upload/update/draw interface (renderer), host scene + camera PODs (scene, camera), and the
one-process-per-GPU tile partition with an RCCL gather (distributed).
"""
import os as _os

# The context keeps three frame kernels in flight on separate HIP streams (plus its main stream); ROCm maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin, and once torch / RCCL have created
# theirs two of ours would share a queue and serialise (measured -10 %).  Must be set before HIP initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import camera, scene  # noqa: F401,E402
from .camera import Camera  # noqa: F401,E402
from .native import NativeError, lib_path, load  # noqa: F401,E402
from .renderer import RVPT, RenderSettings  # noqa: F401,E402

__all__ = ["RVPT", "RenderSettings", "Camera", "scene", "camera", "NativeError", "load", "lib_path"]
