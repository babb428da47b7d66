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


def lib(fused=False) -> C.CDLL:
    """fused: False = no contraction, True = the build's contraction rule; or a variant name: "ieee" / "fused_ieee" (the one
    OpFDiv(OpDot, OpDot) as C's `/`), "libm" (a different admissible driver: libm sin/cos/tan, plain dot and normalize)."""
    if fused in _LIBS:
        return _LIBS[fused]
    name = {False: "libref_spv.so", True: "libref_spv_fused.so"}.get(fused) or f"libref_spv_{fused}.so"
    path = _REF / name
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


# ---- single functions of the module (ref_spv_fn_*): values that never reach a pixel, e.g. the barycentrics ----

class _Vec3(C.Structure):
    _fields_ = [("v", C.c_float * 3)]


class _Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3)]


class _Isect(C.Structure):  # Isect { t, pos, normal, uv, Material_new { type, base_color, emissive, ior } } — 17 words
    _fields_ = [("t", C.c_float), ("pos", C.c_float * 3), ("normal", C.c_float * 3), ("uv", C.c_float * 2), ("mat_type", C.c_int32),
                ("base", C.c_float * 3), ("emissive", C.c_float * 3), ("ior", C.c_float)]


class _Bindings(C.Structure):
    _fields_ = [("binding", C.c_void_p * 8), ("length", C.c_uint32 * 8)]


def _fl(fused):
    L = lib(fused)
    if not getattr(L, "_fn_ready", False):
        bp = C.POINTER(_Bindings)
        fp = C.POINTER(C.c_float)
        L.ref_spv_fn_intersect_triangle_fast.restype = C.c_bool
        L.ref_spv_fn_intersect_triangle_fast.argtypes = [bp, C.POINTER(_Ray), C.POINTER(_Vec3), C.POINTER(_Vec3), C.POINTER(_Vec3), fp, fp, C.POINTER(_Isect)]
        L.ref_spv_fn_intersect_aabb.restype = C.c_bool
        L.ref_spv_fn_intersect_aabb.argtypes = [bp, C.POINTER(_Ray), C.POINTER(_Vec3), C.POINTER(_Vec3), fp, fp]
        L.ref_spv_fn_frensel_reflectance.restype = C.c_float
        L.ref_spv_fn_frensel_reflectance.argtypes = [bp, fp, fp, fp]
        L.ref_spv_fn_map_uniform_sphere.restype = _Vec3
        L.ref_spv_fn_map_uniform_sphere.argtypes = [bp, fp, fp]
        for name in ("camera_pinhole_ray", "camera_ortho_ray", "camera_spherical_ray"):
            f = getattr(L, "ref_spv_fn_" + name)
            f.restype = _Ray
            f.argtypes = [bp, fp, fp]
        L.ref_spv_fn_rand.restype = C.c_float
        L.ref_spv_fn_rand.argtypes = [bp]
        L.ref_spv_fn_wang_hash.restype = C.c_uint32
        L.ref_spv_fn_wang_hash.argtypes = [bp, C.POINTER(C.c_uint32)]
        L.ref_spv_fn_distance_triangle.restype = C.c_float
        L.ref_spv_fn_distance_triangle.argtypes = [bp] + [C.POINTER(_Vec3)] * 4
        L.ref_spv_rand_stream.argtypes = [bp, C.c_uint32, C.c_uint32, C.c_void_p]
        L._fn_ready = True
    return L


def _v3(a):
    return _Vec3((C.c_float * 3)(*[float(x) for x in a]))


def fn_intersect_triangle_fast(org, dirv, v0, v1, v2, mint=0.0, maxt=float("inf"), fused=False):
    """intersect_triangle_fast of the module.  Returns (accept, t, uv[2], normal[3], pos[3]) as float32."""
    L = _fl(fused)
    b = _Bindings()
    ray = _Ray((C.c_float * 3)(*map(float, org)), (C.c_float * 3)(*map(float, dirv)))
    info = _Isect()
    a, c, d = _v3(v0), _v3(v1), _v3(v2)
    lo, hi = C.c_float(mint), C.c_float(maxt)
    acc = L.ref_spv_fn_intersect_triangle_fast(C.byref(b), C.byref(ray), C.byref(a), C.byref(c), C.byref(d), C.byref(lo), C.byref(hi), C.byref(info))
    return bool(acc), np.float32(info.t), np.array(info.uv[:], np.float32), np.array(info.normal[:], np.float32), np.array(info.pos[:], np.float32)


def fn_intersect_aabb(org, dirv, bmin, bmax, mint=0.0, maxt=float("inf"), fused=False):
    L = _fl(fused)
    b = _Bindings()
    ray = _Ray((C.c_float * 3)(*map(float, org)), (C.c_float * 3)(*map(float, dirv)))
    lo, hi = C.c_float(mint), C.c_float(maxt)
    mn, mx = _v3(bmin), _v3(bmax)
    return bool(L.ref_spv_fn_intersect_aabb(C.byref(b), C.byref(ray), C.byref(mn), C.byref(mx), C.byref(lo), C.byref(hi)))


def fn_fresnel(cos_in, cos_out, eta, fused=False):
    L = _fl(fused)
    b = _Bindings()
    a, c, e = C.c_float(cos_in), C.c_float(cos_out), C.c_float(eta)
    return np.float32(L.ref_spv_fn_frensel_reflectance(C.byref(b), C.byref(a), C.byref(c), C.byref(e)))


def fn_map_uniform_sphere(u, v, fused=False):
    L = _fl(fused)
    b = _Bindings()
    a, c = C.c_float(u), C.c_float(v)
    return np.array(L.ref_spv_fn_map_uniform_sphere(C.byref(b), C.byref(a), C.byref(c)).v[:], np.float32)


def fn_camera_ray(kind, camera, x, y, fused=False):
    """kind: 'pinhole' | 'ortho' | 'spherical' (camera.glsl:29-99).  Returns (origin[3], direction[3])."""
    L = _fl(fused)
    cam = np.ascontiguousarray(camera, dtype=np.float32).reshape(20)
    b = _Bindings()
    b.binding[4] = cam.ctypes.data
    a, c = C.c_float(x), C.c_float(y)
    r = getattr(L, f"ref_spv_fn_camera_{kind}_ray")(C.byref(b), C.byref(a), C.byref(c))
    return np.array(r.o[:], np.float32), np.array(r.d[:], np.float32)


def fn_rand_stream(state, n, fused=False):
    """n consecutive draws of the module's rand() starting from rng_state = state (util.glsl:38-50)."""
    L = _fl(fused)
    b = _Bindings()
    out = np.zeros(n, np.float32)
    L.ref_spv_rand_stream(C.byref(b), int(state) & 0xFFFFFFFF, n, out.ctypes.data_as(C.c_void_p))
    return out


def fn_wang_hash(seed, fused=False):
    L = _fl(fused)
    b = _Bindings()
    s = C.c_uint32(int(seed) & 0xFFFFFFFF)
    return int(L.ref_spv_fn_wang_hash(C.byref(b), C.byref(s)))


def fn_distance_triangle(p, a, b_, c, fused=False):
    L = _fl(fused)
    b = _Bindings()
    pp, aa, bb, cc = _v3(p), _v3(a), _v3(b_), _v3(c)
    return np.float32(L.ref_spv_fn_distance_triangle(C.byref(b), C.byref(pp), C.byref(aa), C.byref(bb), C.byref(cc)))
