"""ctypes front-end of oracle/_ref/libref_spv*.so — the reference's compiled shader translated to C (tools/spv2c.py).

TEST INFRASTRUCTURE, authoring container only: the library is built from /root/reference's compute_pass.comp.spv by
oracle/ref_spv/Makefile into the git- and gpurun-ignored oracle/_ref/.  tools/make_ref_golden.py turns its outputs into
the committed fixtures tests/golden/ref_spv_*.npz; tests/test_ref_spv.py re-checks them when the library is present.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_REF = _HERE.parent / "_ref"
_LIBS = {}


def available() -> bool:
    return Path("/root/reference/assets/shaders/compute_pass.comp.spv").exists() or (_REF / "libref_spv.so").exists()


def build() -> None:
    """Run the recipe (needs /root/reference)."""
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)


def lib(fused: bool = False) -> C.CDLL:
    if fused in _LIBS:
        return _LIBS[fused]
    path = _REF / ("libref_spv_fused.so" if fused else "libref_spv.so")
    if not path.exists():
        build()
    L = C.CDLL(str(path))
    vp, u32 = C.c_void_p, C.c_uint32
    L.ref_spv_render.restype = C.c_int
    L.ref_spv_render.argtypes = [vp, vp, vp, u32, vp, u32, vp, u32, u32, u32, vp, vp, C.c_int]
    L.ref_spv_function_count.restype = u32
    L.ref_spv_instruction_count.restype = u32
    _LIBS[fused] = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def render(settings, camera, nodes, tris, mats, width, height, prev=None, unorm8=False, fused=False):
    """One dispatch of the reference shader.  Returns the result image [H,W,4] float32 (also the new temporal image).

    prev is the temporal accumulation image the shader reads (zeros if None, as a freshly created VkImage is cleared by
    the harness).  With unorm8 the images behave as the reference's rgba8 images (values are k/255).
    """
    L = lib(fused)
    settings = np.ascontiguousarray(settings)
    assert settings.nbytes == 40
    camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(20)
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
    mats = np.ascontiguousarray(mats, dtype=np.float32).reshape(-1, 12)
    nodes = np.ascontiguousarray(nodes)
    temporal = np.zeros((height, width, 4), np.float32) if prev is None else np.array(prev, dtype=np.float32, copy=True).reshape(height, width, 4)
    result = np.zeros((height, width, 4), np.float32)
    rc = L.ref_spv_render(_p(settings), _p(camera), _p(nodes), nodes.nbytes // 32, _p(tris), tris.shape[0], _p(mats),
                          mats.shape[0], width, height, _p(temporal), _p(result), int(unorm8))
    if rc != 0:
        raise RuntimeError(f"ref_spv_render failed: {rc}")
    assert np.array_equal(temporal.view(np.uint32), result.view(np.uint32))  # the shader stores the same texel to both
    return result
