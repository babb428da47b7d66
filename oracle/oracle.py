"""ctypes front-end of the CPU oracle (oracle/rvpt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  The product package (rvpt_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

TRAVERSAL_BVH = 0
TRAVERSAL_BRUTE = 1
TRAVERSAL_BVH_ORDERED = 2  # build-defined: nearer child first (the reference's TODO, intersection.glsl:405)


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (both variants).  Building the checker is not using it."""
    out = _HERE / "build"
    names = ("liboracle.so", "liboracle_nofma.so", "liboracle_unfused.so")
    src = (_HERE / "rvpt_oracle.c").stat().st_mtime
    if force or any(not (out / n).exists() or (out / n).stat().st_mtime < src for n in names):
        subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)


def _host_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split()
                    return "fma" in fl and "avx2" in fl
    except OSError:
        pass
    return False


_UNFUSED = None


def lib(unfused: bool = False) -> C.CDLL:
    """The oracle library; unfused=True is the -DORACLE_UNFUSED build (no shader-level FMA contraction)."""
    global _LIB, _UNFUSED
    if unfused and _UNFUSED is not None:
        return _UNFUSED
    if not unfused and _LIB is not None:
        return _LIB
    name = "liboracle_unfused.so" if unfused else ("liboracle.so" if _host_has_fma() else "liboracle_nofma.so")
    path = _HERE / "build" / name
    if not path.exists():
        build()
    L = C.CDLL(str(path))
    vp, sz, u32, i32, f32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
    L.oracle_abi_version.restype = i32
    L.oracle_render.restype = i32
    L.oracle_render.argtypes = [vp, vp, vp, sz, vp, sz, vp, sz, u32, u32, i32, vp, vp, u32, u32, vp]
    L.oracle_prepare.argtypes = [vp, sz, vp]
    L.oracle_render_prepared.restype = i32
    L.oracle_render_prepared.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, sz, u32, u32, i32, vp, vp, u32, u32, vp]
    L.oracle_quantize_rgba8.argtypes = [vp, vp, sz]
    L.oracle_dequantize_rgba8.argtypes = [vp, vp, sz]
    L.oracle_wang_hash.restype = u32
    L.oracle_wang_hash.argtypes = [u32]
    L.oracle_rand_stream.argtypes = [u32, u32, sz, vp, vp]
    L.oracle_sincos.argtypes = [f32, vp, vp]
    L.oracle_tan.restype = f32
    L.oracle_tan.argtypes = [f32]
    L.oracle_sphere_point.argtypes = [f32, f32, vp]
    L.oracle_camera_ray.argtypes = [i32, vp, f32, f32, vp, vp]
    L.oracle_distance_triangle.restype = f32
    L.oracle_distance_triangle.argtypes = [vp, vp, vp, vp]
    L.oracle_set_threads.restype = i32
    L.oracle_set_threads.argtypes = [i32]
    L.oracle_div_dots.restype = f32
    L.oracle_div_dots.argtypes = [f32, f32]
    L.oracle_fresnel.restype = f32
    L.oracle_fresnel.argtypes = [f32, f32, f32]
    L.oracle_tri_test.restype = i32
    L.oracle_tri_test.argtypes = [vp, vp, vp, f32, f32, vp]
    L.oracle_aabb_test.restype = i32
    L.oracle_aabb_test.argtypes = [vp, vp, vp, vp, f32, f32]
    L.oracle_pinhole_ray.argtypes = [vp, f32, f32, vp, vp]
    L.oracle_closest_hit.restype = C.c_long
    L.oracle_closest_hit.argtypes = [vp, sz, vp, sz, i32, vp, vp, vp]
    if unfused:
        _UNFUSED = L
    else:
        _LIB = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c32(a, dtype=np.float32):
    return np.ascontiguousarray(a, dtype=dtype)


def settings_bytes(max_bounces=8, aa=1, current_frame=0, camera_mode=0, modes=(9, 9, 9, 9), split=(0.5, 0.5)):
    """40-byte RenderSettings block (rvpt.h:77-89)."""
    s = np.zeros(10, dtype=np.int32)
    s[0], s[1] = max_bounces, aa
    s.view(np.uint32)[2] = current_frame
    s[3] = camera_mode
    s[4:8] = modes
    s.view(np.float32)[8:10] = split
    return s


def render(settings, camera, nodes, tris, mats, width, height, traversal, prev=None, y0=0, y1=None, threads=None,
           unfused=False):
    """One frame over rows [y0, y1).  Returns (image[H,W,4] float32, stats[2] uint64).

    settings: int32[10] block from settings_bytes(); camera: float32[20]; nodes: structured/bytes
    array of 32-byte nodes or None; tris: float32[N,16]; mats: float32[M,12].
    """
    L = lib(unfused)
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    settings = np.ascontiguousarray(settings)
    camera = _c32(camera).reshape(20)
    tris = _c32(tris).reshape(-1, 16)
    mats = _c32(mats).reshape(-1, 12)
    n_nodes = 0
    if nodes is not None:
        nodes = np.ascontiguousarray(nodes)
        n_nodes = nodes.nbytes // 32
    out = np.zeros((height, width, 4), dtype=np.float32)
    stats = np.zeros(2, dtype=np.uint64)
    if prev is not None:
        prev = _c32(prev).reshape(height, width, 4)
    rc = L.oracle_render(_p(settings), _p(camera), _p(nodes), n_nodes, _p(tris), tris.shape[0], _p(mats),
                         mats.shape[0], width, height, traversal, _p(prev), _p(out), y0,
                         height if y1 is None else y1, _p(stats))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return out, stats


def usable_cores() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (containers)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, min(n, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_threads(n: int) -> int:
    """OpenMP threads of the render calls; returns the previous setting."""
    return int(lib().oracle_set_threads(int(n)))


class PreparedScene:
    """Scene arrays + the per-triangle terms prepared once (oracle_prepare), for timing loops over many frames."""

    def __init__(self, nodes, tris, mats):
        self.tris = _c32(tris).reshape(-1, 16)
        self.mats = _c32(mats).reshape(-1, 12)
        self.nodes = None if nodes is None else np.ascontiguousarray(nodes)
        self.n_nodes = 0 if self.nodes is None else self.nodes.nbytes // 32
        self.prep = np.zeros((max(self.tris.shape[0], 1), 16), dtype=np.float32)
        lib().oracle_prepare(_p(self.tris), self.tris.shape[0], _p(self.prep))

    def render(self, settings, camera, width, height, traversal, out=None, y0=0, y1=None):
        settings = np.ascontiguousarray(settings)
        camera = _c32(camera).reshape(20)
        if out is None:
            out = np.zeros((height, width, 4), dtype=np.float32)
        rc = lib().oracle_render_prepared(_p(settings), _p(camera), _p(self.nodes), self.n_nodes, _p(self.tris), _p(self.prep),
                                          self.tris.shape[0], _p(self.mats), self.mats.shape[0], width, height, traversal, None,
                                          _p(out), y0, height if y1 is None else y1, None)
        if rc != 0:
            raise RuntimeError(f"oracle_render_prepared failed: {rc}")
        return out


def quantize_rgba8(img):
    img = _c32(img)
    out = np.zeros(img.shape, dtype=np.uint8)
    lib().oracle_quantize_rgba8(_p(img), _p(out), img.size)
    return out


def dequantize_rgba8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros(img.shape, dtype=np.float32)
    lib().oracle_dequantize_rgba8(_p(img), _p(out), img.size)
    return out


def wang_hash(seed: int) -> int:
    return int(lib().oracle_wang_hash(seed & 0xFFFFFFFF))


def rand_stream(p_idx: int, frame: int, n: int):
    out = np.zeros(n, dtype=np.float32)
    st = np.zeros(n, dtype=np.uint32)
    lib().oracle_rand_stream(p_idx, frame, n, _p(out), _p(st))
    return out, st


def sincos(x: float):
    s = np.zeros(1, np.float32)
    c = np.zeros(1, np.float32)
    lib().oracle_sincos(float(x), _p(s), _p(c))
    return float(s[0]), float(c[0])


def tan(x: float) -> float:
    return float(lib().oracle_tan(float(x)))


def sphere_point(u: float, v: float, unfused=False):
    o = np.zeros(3, np.float32)
    lib(unfused).oracle_sphere_point(float(u), float(v), _p(o))
    return o


def div_dots(a, b):
    """The ray/plane quotient of the arithmetic specification (o_div_dots) element-wise on float32 arrays."""
    a, b = np.broadcast_arrays(_c32(a), _c32(b))
    f = lib().oracle_div_dots
    return np.array([f(float(x), float(y)) for x, y in zip(a.ravel(), b.ravel())], dtype=np.float32).reshape(a.shape)


def fresnel(cos_in, cos_out, eta, unfused=False) -> float:
    return float(lib(unfused).oracle_fresnel(float(cos_in), float(cos_out), float(eta)))


def tri_test(org, dirv, tri, mint=0.0, maxt=float("inf"), unfused=False):
    org, dirv, tri = _c32(org), _c32(dirv), _c32(tri).reshape(16)
    tuv = np.zeros(3, np.float32)
    acc = lib(unfused).oracle_tri_test(_p(org), _p(dirv), _p(tri), mint, maxt, _p(tuv))
    return bool(acc), tuv


def aabb_test(org, dirv, bmin, bmax, mint=0.0, maxt=float("inf"), unfused=False) -> bool:
    org, dirv, bmin, bmax = _c32(org), _c32(dirv), _c32(bmin), _c32(bmax)
    return bool(lib(unfused).oracle_aabb_test(_p(org), _p(dirv), _p(bmin), _p(bmax), mint, maxt))


def pinhole_ray(camera, x, y):
    camera = _c32(camera).reshape(20)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib().oracle_pinhole_ray(_p(camera), float(x), float(y), _p(o), _p(d))
    return o, d


def camera_ray(mode, camera, x, y, unfused=False):
    camera = _c32(camera).reshape(20)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib(unfused).oracle_camera_ray(int(mode), _p(camera), float(x), float(y), _p(o), _p(d))
    return o, d


def distance_triangle(p, a, b, c, unfused=False) -> float:
    p, a, b, c = _c32(p), _c32(a), _c32(b), _c32(c)
    return np.float32(lib(unfused).oracle_distance_triangle(_p(p), _p(a), _p(b), _p(c)))


def prepare(tris, unfused=False):
    """oracle_prepare: float32[n,16] records (v0, n, e0, e1, a00, a01, a11, inv_det)."""
    tris = _c32(tris).reshape(-1, 16)
    out = np.zeros((tris.shape[0], 16), np.float32)
    lib(unfused).oracle_prepare(_p(tris), tris.shape[0], _p(out))
    return out


def closest_hit(nodes, tris, traversal, org, dirv):
    tris = _c32(tris).reshape(-1, 16)
    n_nodes = 0
    if nodes is not None:
        nodes = np.ascontiguousarray(nodes)
        n_nodes = nodes.nbytes // 32
    org, dirv = _c32(org), _c32(dirv)
    t = np.zeros(1, np.float32)
    h = lib().oracle_closest_hit(_p(nodes), n_nodes, _p(tris), tris.shape[0], traversal, _p(org), _p(dirv), _p(t))
    return int(h), float(t[0])
