"""ctypes bindings of the C ABI (include/rvpt_hip.h).

There is no fallback: if librvpt_hip.so is missing or a call fails, NativeError is raised.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
_LIB = None
_LAB = None

# include/rvpt_hip.h constants
ABI_VERSION = 8
BUILD_LAB, BUILD_DEBUG_CHECKS = 0x1, 0x2  # rvpt_hip_build_flags
MAX_FRAMES_PER_DISPATCH = 64
TRAVERSAL_BRUTE, TRAVERSAL_BVH, TRAVERSAL_BVH_ORDERED = 0x0, 0x1, 0x2
COUNT_SEGMENTS, KERNEL_SIMPLE, TIMING, ACCUM_UNORM8 = 0x4, 0x8, 0x10, 0x20
BVH_PER_LANE = 0x400  # BVH contexts: no camera packets, every segment walks the tree per lane (rounds 1-3's kernel)
BRUTE_MIXED_PACKETS = 0x200  # brute-force contexts: round 2's mixed-packet frame kernel instead of the packet kernel (rvpt_packets.hip)
FORMAT_RGBA32F, FORMAT_RGBA8_UNORM = 0, 1
TILE = 16
TILE_SHIFT = 3  # RVPT_HIP_TILE_SHIFT: every row of the tile grid is rotated by this many more tiles than the one above (tile ownership)
ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_SIZE, ERR_COMM = -1, -2, -3, -4, -5, -6

# the C ABI of include/rvpt_hip.h: what librvpt_hip.so exports, all of it and nothing else (tests/test_abi_exports.py)
EXPORTS = [
    "rvpt_hip_abi_version", "rvpt_hip_build_flags", "rvpt_hip_device_count", "rvpt_hip_create", "rvpt_hip_destroy",
    "rvpt_hip_upload_scene", "rvpt_hip_set_frame", "rvpt_hip_dispatch", "rvpt_hip_dispatch_frames", "rvpt_hip_wait", "rvpt_hip_wait_for", "rvpt_hip_query",
    "rvpt_hip_read", "rvpt_hip_tile_buffer", "rvpt_hip_untile", "rvpt_hip_write_accum", "rvpt_hip_get_timing",
    "rvpt_hip_reset_timing", "rvpt_hip_get_stats", "rvpt_hip_get_launch_info", "rvpt_hip_get_cull_info", "rvpt_hip_last_error", "rvpt_bvh_build",
    "rvpt_hip_comm_unique_id", "rvpt_hip_comm_init", "rvpt_hip_comm_init_all", "rvpt_hip_comm_info", "rvpt_hip_gather", "rvpt_hip_comm_barrier", "rvpt_hip_comm_destroy",
]
# ... and what the laboratory build librvpt_hip_debug.so adds (include/rvpt_hip_lab.h)
LAB_EXPORTS = [
    "rvpt_hip_selftest_div", "rvpt_hip_selftest_rcp", "rvpt_hip_selftest_pretest", "rvpt_hip_selftest_camera_rects", "rvpt_hip_selftest_bounce_cull",
    "rvpt_hip_selftest_fast_div", "rvpt_camera_rects", "rvpt_bounce_rows", "rvpt_bounce_leaf_boxes", "rvpt_bvh_wide_form", "rvpt_bvh_quant_form", "rvpt_claim_order",
]


class NativeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rvpt_hip error {code}: {message}")
        self.code = code


def lib_path(lab: bool = False) -> Path:
    """In-tree library: librvpt_hip.so, or the laboratory build librvpt_hip_debug.so (include/rvpt_hip_lab.h).  RVPT_HIP_LIB overrides the release path
    (kernel experiments: tools/archive/exp_variants.py), RVPT_HIP_LAB_LIB the laboratory one."""
    import os
    override = os.environ.get("RVPT_HIP_LAB_LIB" if lab else "RVPT_HIP_LIB")
    return Path(override) if override else _PKG / ("librvpt_hip_debug.so" if lab else "librvpt_hip.so")


def _bind(L, lab: bool):
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    L.rvpt_hip_abi_version.restype = i32
    L.rvpt_hip_build_flags.restype = u32
    L.rvpt_hip_device_count.argtypes = [C.POINTER(i32)]
    L.rvpt_hip_create.argtypes = [C.POINTER(vp), i32, u32, u32, u32, u32, u32]
    L.rvpt_hip_destroy.argtypes = [vp]
    L.rvpt_hip_destroy.restype = None
    L.rvpt_hip_upload_scene.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.rvpt_hip_set_frame.argtypes = [vp, vp, vp]
    L.rvpt_hip_dispatch.argtypes = [vp]
    L.rvpt_hip_dispatch_frames.argtypes = [vp, C.c_uint32]
    L.rvpt_hip_wait_for.argtypes = [vp, C.c_uint64]
    L.rvpt_hip_wait.argtypes = [vp]
    L.rvpt_hip_query.argtypes = [vp]
    L.rvpt_hip_read.argtypes = [vp, i32, vp, sz]
    L.rvpt_hip_tile_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.rvpt_hip_untile.argtypes = [vp, vp, sz, u32, vp]
    L.rvpt_hip_write_accum.argtypes = [vp, vp, sz]
    L.rvpt_hip_get_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.rvpt_hip_reset_timing.argtypes = [vp]
    L.rvpt_hip_get_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rvpt_hip_get_launch_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.rvpt_hip_get_cull_info.argtypes = [vp, C.POINTER(u32)]
    L.rvpt_hip_last_error.argtypes = [vp]
    L.rvpt_hip_last_error.restype = C.c_char_p
    L.rvpt_bvh_build.argtypes = [vp, sz, vp, C.POINTER(sz), vp]
    L.rvpt_hip_comm_unique_id.argtypes = [vp, sz]
    L.rvpt_hip_comm_init.argtypes = [vp, vp, sz]
    L.rvpt_hip_comm_init_all.argtypes = [C.POINTER(vp), i32]
    L.rvpt_hip_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.rvpt_hip_gather.argtypes = [vp, vp]
    L.rvpt_hip_comm_barrier.argtypes = [vp]
    L.rvpt_hip_comm_destroy.argtypes = [vp]
    names = list(EXPORTS)
    if lab:
        L.rvpt_bvh_wide_form.argtypes = [vp, sz, C.c_uint32, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint32)]
        L.rvpt_bvh_quant_form.argtypes = [vp, sz, C.c_uint32, sz, vp, sz, C.POINTER(sz), vp, C.POINTER(C.c_float)]
        L.rvpt_hip_selftest_div.argtypes = [i32, vp, vp, vp, sz]
        L.rvpt_hip_selftest_rcp.argtypes = [i32, vp]
        L.rvpt_hip_selftest_pretest.argtypes = [i32, vp, vp, vp, vp, sz]
        L.rvpt_camera_rects.argtypes = [vp, sz, vp, u32, u32, vp]
        L.rvpt_bounce_rows.argtypes = [vp, vp, sz, vp, C.POINTER(C.c_double)]
        L.rvpt_bounce_leaf_boxes.argtypes = [vp, sz, vp, C.POINTER(u32), vp]
        L.rvpt_claim_order.argtypes = [u32, u32, vp, vp]
        L.rvpt_hip_selftest_camera_rects.argtypes = [vp, u32, vp, vp, vp]
        L.rvpt_hip_selftest_bounce_cull.argtypes = [vp, u32, vp]
        L.rvpt_hip_selftest_fast_div.argtypes = [u32, vp, vp, sz]
        names += LAB_EXPORTS
    for name in names:
        if name not in ("rvpt_hip_destroy", "rvpt_hip_last_error", "rvpt_hip_build_flags"):
            getattr(L, name).restype = i32
    if L.rvpt_hip_abi_version() != ABI_VERSION:
        raise NativeError(ERR_INVALID, f"ABI version {L.rvpt_hip_abi_version()} != {ABI_VERSION}")
    if lab and not (L.rvpt_hip_build_flags() & BUILD_LAB):
        raise NativeError(ERR_INVALID, "the laboratory library was not built with -DRVPT_HIP_LAB=1 (rvpt_amd.build.build_native_debug)")
    return L


def _open(lab: bool) -> C.CDLL:
    # torch bundles its own ROCm runtime (libamdhip64.so.7 + HSA) and must be the first to load it: if
    # the system copy of the same SONAME gets in first (through this library's DT_NEEDED), torch ends up
    # on a mixed runtime that sees no device.  torch is only plumbing here, but load order matters.
    import torch  # noqa: F401
    path = lib_path(lab)
    if not path.exists():
        raise NativeError(ERR_HIP, f"{path} not found — run `python -m rvpt_amd.build` (hipcc, gfx950); "
                                   "there is no CPU fallback")
    return _bind(C.CDLL(str(path)), lab)


def load() -> C.CDLL:
    """Load librvpt_hip.so (built in-tree by rvpt_amd.build / __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        _LIB = _open(False)
    return _LIB


def load_lab() -> C.CDLL:
    """Load librvpt_hip_debug.so, the laboratory build (include/rvpt_hip_lab.h): the release ABI + selftests, host-side forms, opt-in walks, knobs, internal checks.
    A second library in the same process; contexts of the two never mix."""
    global _LAB
    if _LAB is None:
        _LAB = _open(True)
    return _LAB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc: int, ctx=None, L=None) -> None:
    if rc != 0:
        msg = (L or load()).rvpt_hip_last_error(ctx)
        raise NativeError(rc, msg.decode() if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    _check(load().rvpt_hip_device_count(C.byref(n)))
    return n.value


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """rvpt_hip_comm_unique_id: 128 bytes made by rank 0, to be handed to every rank's Context.comm_init."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load().rvpt_hip_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def comm_init_all(contexts) -> None:
    """rvpt_hip_comm_init_all: one process, one context per GPU, in rank order."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _check(load().rvpt_hip_comm_init_all(arr, len(contexts)), contexts[0]._h)


def selftest_div(a, b, device: int = 0) -> np.ndarray:
    """rvpt_hip_selftest_div: the kernels' ray/plane quotient (div_dots) of two float32 arrays, evaluated on the GPU."""
    a, b = np.broadcast_arrays(np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32))
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    out = np.zeros(a.shape, dtype=np.float32)
    _check(load_lab().rvpt_hip_selftest_div(device, _ptr(a), _ptr(b), _ptr(out), a.size), None, load_lab())
    return out


def selftest_pretest(a, den, closest, device: int = 0) -> np.ndarray:
    """rvpt_hip_selftest_pretest: per element, bit 0 = the division-free pre-test of a camera round lets (a, den, closest) through,
    bit 1 = the quotient t = div_dots(a, den) satisfies 0 < t < closest."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    den = np.ascontiguousarray(den, dtype=np.float32)
    closest = np.ascontiguousarray(closest, dtype=np.float32)
    assert a.shape == den.shape == closest.shape
    out = np.zeros(a.shape, dtype=np.uint8)
    _check(load_lab().rvpt_hip_selftest_pretest(device, _ptr(a), _ptr(den), _ptr(closest), _ptr(out), a.size), None, load_lab())
    return out


def selftest_rcp(device: int = 0) -> np.ndarray:
    """rvpt_hip_selftest_rcp: per exponent, how many binary32 b have a refined v_rcp_f32 != the correctly rounded 1/b."""
    out = np.zeros(256, dtype=np.uint64)
    _check(load_lab().rvpt_hip_selftest_rcp(device, _ptr(out)), None, load_lab())
    return out


def build_bvh(tris: np.ndarray):
    """rvpt_bvh_build: returns (nodes uint8[n_nodes,32] view-able as the node struct, prim_indices uint32[n])."""
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
    n = tris.shape[0]
    nodes = np.zeros((max(2 * n - 1, 1), 8), dtype=np.uint32)
    idx = np.zeros(n, dtype=np.uint32)
    n_nodes = C.c_size_t(0)
    _check(load().rvpt_bvh_build(_ptr(tris), n, _ptr(nodes), C.byref(n_nodes), _ptr(idx)))
    return nodes[: n_nodes.value].copy(), idx


def wide_form(nodes: np.ndarray, head_shift: int):
    """rvpt_bvh_wide_form: the 4-wide regrouping of a binary tree (uint32[n, 8] or NODE_DTYPE records) that BVH contexts walk by default.
    Returns (wide float32[n_wide, 8, 4] — quads minx maxx miny maxy minz maxz head pad; view heads as uint32 —, stack_need)."""
    nodes = np.ascontiguousarray(nodes).view(np.uint32).reshape(-1, 8)
    n = nodes.shape[0]
    out = np.zeros((max(n, 1), 8, 4), dtype=np.float32)
    n_wide, need = C.c_size_t(0), C.c_uint32(0)
    _check(load_lab().rvpt_bvh_wide_form(_ptr(nodes), n, int(head_shift), _ptr(out), out.shape[0], C.byref(n_wide), C.byref(need)), None, load_lab())
    return out[: n_wide.value].copy(), int(need.value)


def quant_form(nodes: np.ndarray, head_shift: int, n_tris: int):
    """rvpt_bvh_quant_form: the 64-byte quantised wide nodes (RVPT_HIP_BVH_QUANT=1) of a binary tree.  Returns (quant uint32[n, 16], leaf boxes
    float32[n_tris, 8], extent); n == 0 when the tree has no quantised form."""
    nodes = np.ascontiguousarray(nodes).view(np.uint32).reshape(-1, 8)
    n = nodes.shape[0]
    out = np.zeros((max(n, 1), 16), dtype=np.uint32)
    boxes = np.zeros((max(int(n_tris), 1), 8), dtype=np.float32)
    n_q, extent = C.c_size_t(0), C.c_float(0.0)
    _check(load_lab().rvpt_bvh_quant_form(_ptr(nodes), n, int(head_shift), int(n_tris), _ptr(out), out.shape[0], C.byref(n_q), _ptr(boxes), C.byref(extent)), None, load_lab())
    return out[: n_q.value].copy(), boxes, float(extent.value)


def fast_div(x: np.ndarray, divisor: int) -> np.ndarray:
    """rvpt_hip_selftest_fast_div: x // divisor through the kernels' multiply-high division (no GPU needed)."""
    x = np.ascontiguousarray(x, dtype=np.uint32)
    q = np.zeros_like(x)
    _check(load_lab().rvpt_hip_selftest_fast_div(int(divisor), _ptr(x), _ptr(q), x.size), None, load_lab())
    return q


def camera_rects(prepared: np.ndarray, camera: np.ndarray, width: int, height: int) -> np.ndarray:
    """rvpt_camera_rects (no GPU needed): the screen rectangles of the packet kernel's camera rounds for prepared records float32[n, 16] and the
    80-byte camera block.  Returns int32[n, 4] = (x0, x1, y0, y1) in units of 16 pixels / 4 rows; x0 > x1 = no block."""
    prepared = np.ascontiguousarray(prepared, dtype=np.float32).reshape(-1, 16)
    camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(20)
    out = np.zeros((prepared.shape[0], 2), dtype=np.uint32)
    _check(load_lab().rvpt_camera_rects(_ptr(prepared), prepared.shape[0], _ptr(camera), int(width), int(height), _ptr(out)), None, load_lab())
    return unpack_rects(out)


def bounce_rows(tris: np.ndarray, prepared: np.ndarray):
    """rvpt_bounce_rows (no GPU needed): the bounce cull's table for reference Triangle records float32[n, 16] and their prepared records float32[n, 16], as
    upload_scene builds it.  Returns (rows uint32[2 n, ceil(n / 32)], scene scale); scale 0 = no table for this scene."""
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
    prepared = np.ascontiguousarray(prepared, dtype=np.float32).reshape(-1, 16)
    n = tris.shape[0]
    rows = np.zeros((2 * n, (n + 31) // 32), dtype=np.uint32)
    scale = C.c_double(0.0)
    _check(load_lab().rvpt_bounce_rows(_ptr(tris), _ptr(prepared), n, _ptr(rows), C.byref(scale)), None, load_lab())
    return rows, float(scale.value)


def claim_order(n_work_frame: int, group_blocks: int = 1):
    """rvpt_claim_order (no GPU needed): (order uint32[n_work_frame / 64], (groups, stride, shift)) — the tile-linear block the b-th block of the packet kernel's
    claim order is, for launches of fewer than four frames; groups == 0: this size keeps the tile-linear order."""
    order = np.zeros(n_work_frame // 64, dtype=np.uint32)
    params = np.zeros(3, dtype=np.uint32)
    _check(load_lab().rvpt_claim_order(n_work_frame, group_blocks, _ptr(order), _ptr(params)), None, load_lab())
    return order, tuple(int(x) for x in params)


def bounce_leaf_boxes(tris: np.ndarray, with_triangles: bool = False):
    """rvpt_bounce_leaf_boxes (no GPU needed): (boxes float32[n_leaves, 8] = lo.xyz, hi.xyz, 0, 0; triangles per leaf) as upload_scene builds them; with_triangles:
    also every triangle's own box float32[n, 8] (the second level)."""
    tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
    per = C.c_uint32(0)
    boxes = np.zeros((max(1, (tris.shape[0] + 3) // 4), 8), dtype=np.float32)  # room for leaves of four
    own = np.zeros((max(1, tris.shape[0]), 8), dtype=np.float32) if with_triangles else None
    _check(load_lab().rvpt_bounce_leaf_boxes(_ptr(tris), tris.shape[0], _ptr(boxes), C.byref(per), _ptr(own)), None, load_lab())
    leaves = boxes[: (tris.shape[0] + per.value - 1) // per.value].copy()
    return (leaves, int(per.value), own[: tris.shape[0]]) if with_triangles else (leaves, int(per.value))


def build_flags(lab: bool = False) -> int:
    """rvpt_hip_build_flags of the release (or laboratory) library: BUILD_LAB | BUILD_DEBUG_CHECKS."""
    return int((load_lab() if lab else load()).rvpt_hip_build_flags())


def unpack_rects(words: np.ndarray) -> np.ndarray:
    words = np.asarray(words, dtype=np.uint32).reshape(-1, 2)
    return np.stack([words[:, 0] & 0xFFFF, words[:, 0] >> 16, words[:, 1] & 0xFFFF, words[:, 1] >> 16], axis=1).astype(np.int32)


NODE_DTYPE = np.dtype([("first", "<u4"), ("count", "<u4"), ("bounds", "<f4", (6,))])


class Context:
    """One rvpt_hip_ctx: one GPU, one image partition."""

    def __init__(self, width: int, height: int, device: int = 0, tile_rank: int = 0, tile_world: int = 1, flags: int = 0, lab=None):
        """lab=True: a context of the laboratory build (include/rvpt_hip_lab.h: selftests, opt-in walks, knobs, internal checks).  lab=None (default): the
        release library, unless RVPT_HIP_LAB=1 is in the environment when the context is made (experiments and tests that turn the laboratory's knobs)."""
        import os
        self.lab = bool(lab) if lab is not None else os.environ.get("RVPT_HIP_LAB") == "1"
        self._L = load_lab() if self.lab else load()
        self._h = C.c_void_p(None)
        self.width, self.height = int(width), int(height)
        self.tile_rank, self.tile_world, self.flags = int(tile_rank), int(tile_world), int(flags)
        _check(self._L.rvpt_hip_create(C.byref(self._h), device, width, height, tile_rank, tile_world, flags), None, self._L)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.rvpt_hip_destroy(self._h)
            self._h = C.c_void_p(None)

    __del__ = close

    def upload_scene(self, nodes, tris, mats) -> None:
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 16)
        mats = np.ascontiguousarray(mats, dtype=np.float32).reshape(-1, 12)
        n_nodes = 0
        if nodes is not None:
            nodes = np.ascontiguousarray(nodes)
            n_nodes = nodes.nbytes // 32
        _check(self._L.rvpt_hip_upload_scene(self._h, _ptr(nodes), n_nodes, _ptr(tris), tris.shape[0], _ptr(mats),
                                             mats.shape[0]), self._h, self._L)

    def set_frame(self, settings: np.ndarray, camera: np.ndarray) -> None:
        settings = np.ascontiguousarray(settings)
        camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(20)
        if settings.nbytes != 40:
            raise NativeError(ERR_INVALID, "settings block must be 40 bytes")
        _check(self._L.rvpt_hip_set_frame(self._h, _ptr(settings), _ptr(camera)), self._h, self._L)

    def set_frame_fast(self, rs, camera: np.ndarray) -> None:
        """set_frame from a RenderSettings object without per-call allocations (the per-frame host loop)."""
        buf = getattr(self, "_rs_buf", None)
        if buf is None:
            buf = self._rs_buf = np.zeros(10, dtype=np.int32)
            self._rs_u32, self._rs_f32 = buf.view(np.uint32), buf.view(np.float32)
            self._rs_ptr = buf.ctypes.data_as(C.c_void_p)
        buf[0], buf[1] = rs.max_bounces, rs.aa
        self._rs_u32[2] = rs.current_frame & 0xFFFFFFFF
        buf[3], buf[4], buf[5], buf[6], buf[7] = (rs.camera_mode, rs.top_left_render_mode, rs.top_right_render_mode,
                                                  rs.bottom_left_render_mode, rs.bottom_right_render_mode)
        self._rs_f32[8], self._rs_f32[9] = rs.split_ratio
        cam = getattr(self, "_cam_last", None)
        if cam is None or cam[0] is not camera:
            c = np.ascontiguousarray(camera, dtype=np.float32).reshape(20)
            cam = self._cam_last = (camera, c, c.ctypes.data_as(C.c_void_p))
        rc = self._L.rvpt_hip_set_frame(self._h, self._rs_ptr, cam[2])
        if rc:
            _check(rc, self._h, self._L)

    def dispatch(self) -> None:
        rc = self._L.rvpt_hip_dispatch(self._h)
        if rc:
            _check(rc, self._h, self._L)

    def dispatch_frames(self, n_frames: int) -> None:
        """n consecutive frames starting at the last set_frame()'s current_frame, as one launch."""
        rc = self._L.rvpt_hip_dispatch_frames(self._h, n_frames)
        if rc:
            _check(rc, self._h, self._L)

    def wait(self) -> None:
        _check(self._L.rvpt_hip_wait(self._h), self._h, self._L)

    def wait_for(self, timeout_s: float) -> bool:
        """True when everything dispatched so far has finished within timeout_s, False if still pending."""
        rc = self._L.rvpt_hip_wait_for(self._h, int(timeout_s * 1e9))
        if rc < 0:
            _check(rc, self._h, self._L)
        return rc == 0

    def query(self) -> bool:
        """True while work is pending."""
        rc = self._L.rvpt_hip_query(self._h)
        if rc < 0:
            _check(rc, self._h, self._L)
        return rc == 1

    def read(self, fmt: int = FORMAT_RGBA32F) -> np.ndarray:
        dt = np.float32 if fmt == FORMAT_RGBA32F else np.uint8
        out = np.empty((self.height, self.width, 4), dtype=dt)
        _check(self._L.rvpt_hip_read(self._h, fmt, _ptr(out), out.nbytes), self._h, self._L)
        return out

    def write_accum(self, img: np.ndarray) -> None:
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(self.height, self.width, 4)
        _check(self._L.rvpt_hip_write_accum(self._h, _ptr(img), img.nbytes), self._h, self._L)

    def tile_buffer(self):
        """(device_ptr, bytes, max_tile_bytes) of this rank's tile-linear accumulator."""
        p, b, m = C.c_void_p(None), C.c_size_t(0), C.c_size_t(0)
        _check(self._L.rvpt_hip_tile_buffer(self._h, C.byref(p), C.byref(b), C.byref(m)), self._h, self._L)
        return p.value, b.value, m.value

    def comm_init(self, unique_id: bytes) -> None:
        """rvpt_hip_comm_init: join the RCCL communicator of this image's tile_world ranks (rank = tile_rank)."""
        _check(self._L.rvpt_hip_comm_init(self._h, unique_id, len(unique_id)), self._h, self._L)

    def comm_info(self):
        """rvpt_hip_comm_info: (ranks, this rank, RCCL version) as RCCL itself reports them for the context's communicator."""
        n, r, v = C.c_int(0), C.c_int(-1), C.c_int(0)
        _check(self._L.rvpt_hip_comm_info(self._h, C.byref(n), C.byref(r), C.byref(v)), self._h, self._L)
        return n.value, r.value, v.value

    def comm_destroy(self) -> None:
        """rvpt_hip_comm_destroy: leave the communicator (reads become local again)."""
        _check(self._L.rvpt_hip_comm_destroy(self._h), self._h, self._L)

    def comm_barrier(self) -> None:
        """rvpt_hip_comm_barrier (collective): this rank's work has finished, then a one-float all-reduce on the communicator."""
        _check(self._L.rvpt_hip_comm_barrier(self._h), self._h, self._L)

    def gather(self, dst_ptr) -> None:
        """rvpt_hip_gather (collective): rank 0 passes a device pointer to width*height*16 bytes, the others None."""
        _check(self._L.rvpt_hip_gather(self._h, C.c_void_p(dst_ptr) if dst_ptr else None), self._h, self._L)

    def untile(self, gathered_ptr: int, slot_bytes: int, n_ranks: int, dst_ptr: int) -> None:
        _check(self._L.rvpt_hip_untile(self._h, C.c_void_p(gathered_ptr), slot_bytes, n_ranks, C.c_void_p(dst_ptr)), self._h, self._L)

    def timing(self):
        """(last_ms, sum_ms, n_dispatches) of the frame kernel (needs the TIMING flag)."""
        last, tot, n = C.c_float(0), C.c_double(0), C.c_uint64(0)
        _check(self._L.rvpt_hip_get_timing(self._h, C.byref(last), C.byref(tot), C.byref(n)), self._h, self._L)
        return last.value, tot.value, n.value

    def reset_timing(self) -> None:
        _check(self._L.rvpt_hip_reset_timing(self._h), self._h, self._L)

    def launch_info(self):
        """(work-groups, LDS bytes per work-group, kernel variant, frames in flight) of the last dispatch."""
        g, l, v, f = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(self._L.rvpt_hip_get_launch_info(self._h, C.byref(g), C.byref(l), C.byref(v), C.byref(f)), self._h, self._L)
        return g.value, l.value, v.value, f.value

    def cull_info(self) -> int:
        """rvpt_hip_get_cull_info of the last launch: bit 0 screen rectangles, bit 1 bounce table, bit 2 camera rounds aligned to 16 x 4 blocks, bit 4 leaf boxes,
        bit 5 interleaved claim order, bit 6 the kernel instance without the uncull'd walks."""
        f = C.c_uint32(0)
        _check(self._L.rvpt_hip_get_cull_info(self._h, C.byref(f)), self._h, self._L)
        return f.value

    def selftest_camera_rects(self, n_samples: int = 1, n_tris: int = 0):
        """rvpt_hip_selftest_camera_rects on this context's scene / camera / image size.  Returns (counts, prepared, rects): counts = (accepted pairs,
        accepted pairs outside their rectangle — the claim is 0 —, (block, triangle) pairs whose rectangle holds the block, all such pairs); with
        n_tris > 0 also the device's prepared records float32[n_tris, 16] and rectangles int32[n_tris, 4]."""
        out = (C.c_uint64 * 4)()
        prep = np.zeros((n_tris, 16), dtype=np.float32) if n_tris else None
        rects = np.zeros((n_tris, 2), dtype=np.uint32) if n_tris else None
        _check(self._L.rvpt_hip_selftest_camera_rects(self._h, int(n_samples), out, _ptr(prep), _ptr(rects)), self._h, self._L)
        return tuple(int(x) for x in out), prep, (unpack_rects(rects) if n_tris else None)

    def selftest_bounce_cull(self, n_samples: int = 1):
        """rvpt_hip_selftest_bounce_cull: (accepted pairs on segments that leave a triangle, those the bounce cull's table excludes — the claim is 0 —, bits set in
        the table, bits in the table, accepted pairs whose ray fails its triangle's leaf box — the claim is 0 —, (segment, leaf box) pairs tested, passed, 0)."""
        out = (C.c_uint64 * 8)()
        _check(self._L.rvpt_hip_selftest_bounce_cull(self._h, int(n_samples), out), self._h, self._L)
        return tuple(int(x) for x in out)

    def stats(self):
        """(segments, samples) traced since create / reset_timing (needs COUNT_SEGMENTS)."""
        s = (C.c_uint64 * 2)()
        _check(self._L.rvpt_hip_get_stats(self._h, s), self._h, self._L)
        return int(s[0]), int(s[1])
