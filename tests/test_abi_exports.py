"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the entry points
include/rvpt_hip.h declares; POD layouts match the reference's GPU structs; the host BVH builder (no GPU
needed) produces a valid tree in the reference node layout."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from rvpt_amd import build, native
    build.build_native()
    return native.load()


def declared_functions(header="rvpt_hip.h"):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rvpt_(?:hip|bvh|camera|bounce|claim)_[a-z_]+)\s*\(", text)))


def exported_functions(path):
    """every C-linkage function a shared object defines (kernels and C++ helpers are mangled: _Z...)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], check=True, capture_output=True, text=True).stdout
    return sorted(line.split()[2] for line in out.splitlines() if len(line.split()) == 3 and line.split()[1] == "T" and not line.split()[2].startswith("_"))


RELEASE_ABI = [
    "rvpt_bvh_build", "rvpt_hip_abi_version", "rvpt_hip_build_flags", "rvpt_hip_comm_barrier", "rvpt_hip_comm_destroy", "rvpt_hip_comm_info", "rvpt_hip_comm_init",
    "rvpt_hip_comm_init_all", "rvpt_hip_comm_unique_id", "rvpt_hip_create", "rvpt_hip_destroy", "rvpt_hip_device_count", "rvpt_hip_dispatch",
    "rvpt_hip_dispatch_frames", "rvpt_hip_gather", "rvpt_hip_get_cull_info", "rvpt_hip_get_launch_info", "rvpt_hip_get_stats", "rvpt_hip_get_timing",
    "rvpt_hip_last_error", "rvpt_hip_query", "rvpt_hip_read", "rvpt_hip_reset_timing", "rvpt_hip_set_frame", "rvpt_hip_tile_buffer", "rvpt_hip_untile",
    "rvpt_hip_upload_scene", "rvpt_hip_wait", "rvpt_hip_wait_for", "rvpt_hip_write_accum",
]
LAB_ABI = [
    "rvpt_bounce_leaf_boxes", "rvpt_bounce_rows", "rvpt_bvh_quant_form", "rvpt_bvh_wide_form", "rvpt_camera_rects", "rvpt_claim_order", "rvpt_hip_selftest_bounce_cull", "rvpt_hip_selftest_camera_rects",
    "rvpt_hip_selftest_div", "rvpt_hip_selftest_fast_div", "rvpt_hip_selftest_pretest", "rvpt_hip_selftest_rcp",
]


def test_the_release_library_exports_the_c_abi_and_nothing_else(lib):
    """include/rvpt_hip.h == native.EXPORTS == what librvpt_hip.so defines == the 30 names pinned here (the laboratory is not in the shipped ABI)."""
    from rvpt_amd import build, native
    names = declared_functions()
    assert names == RELEASE_ABI
    assert sorted(native.EXPORTS) == names
    assert exported_functions(build.LIB_PATH) == names
    for n in names:
        assert hasattr(lib, n), n
    assert not any(hasattr(lib, n) for n in LAB_ABI)
    assert lib.rvpt_hip_build_flags() == 0


def test_the_laboratory_library_adds_the_diagnostics(lib):
    """include/rvpt_hip_lab.h == native.LAB_EXPORTS == what librvpt_hip_debug.so defines beside the release ABI."""
    from rvpt_amd import build, native
    build.build_native_debug()
    assert declared_functions("rvpt_hip_lab.h") == LAB_ABI == sorted(native.LAB_EXPORTS)
    assert exported_functions(build.DEBUG_LIB_PATH) == sorted(RELEASE_ABI + LAB_ABI)
    lab = native.load_lab()
    assert lab.rvpt_hip_build_flags() == native.BUILD_LAB | native.BUILD_DEBUG_CHECKS and lab.rvpt_hip_abi_version() == native.ABI_VERSION


def test_the_release_library_reads_no_laboratory_knob():
    """The tuning knobs the sweeps found flat are constants in the release build: only the names include/rvpt_hip_lab.h lists for it appear in its strings."""
    from rvpt_amd import build
    data = build.LIB_PATH.read_bytes()
    found = sorted(set(m.group(0).decode() for m in re.finditer(rb"RVPT_(?:HIP|BVH)_[A-Z0-9_]{3,}", data)))
    allowed = {"RVPT_HIP_QUIET", "RVPT_HIP_DEBUG", "RVPT_HIP_FRAMES_IN_FLIGHT", "RVPT_HIP_NO_OVERLAP", "RVPT_HIP_PACKETS_CULL", "RVPT_HIP_PACKETS_BOUNCE_CULL", "RVPT_HIP_PACKETS_BOX_CULL", "RVPT_HIP_PACKETS_INTERLEAVE",
               "RVPT_HIP_COMM_TIMEOUT_S", "RVPT_BVH_THREADS", "RVPT_BVH_TRAVERSAL_COST"}
    macros = set(re.findall(r"#define (RVPT_HIP_[A-Z0-9_]+)", (ROOT / "include" / "rvpt_hip.h").read_text()))  # flag names in error messages
    assert set(found) - macros <= allowed, sorted(set(found) - macros - allowed)


def test_abi_version(lib):
    assert lib.rvpt_hip_abi_version() == 8


def test_header_struct_sizes_match_reference_layouts():
    text = (ROOT / "include" / "rvpt_hip.h").read_text()
    # sizes asserted at compile time in rvpt_abi.hip; here: the header is self-contained C
    import subprocess, tempfile
    src = '#include "rvpt_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(rvpt_triangle), sizeof(rvpt_bvh_node), sizeof(rvpt_material), sizeof(rvpt_render_settings), sizeof(rvpt_camera_data));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "t.c").write_text(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(Path(d) / "t.c"), "-o", str(Path(d) / "t")], check=True)
        out = subprocess.run([str(Path(d) / "t")], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["64", "32", "48", "40", "80"]
    assert "rvpt_triangle" in text


def test_header_constants_agree_with_the_bindings_and_the_kernels():
    """Flag bits, the tile size and the ownership rotation exist three times (include/rvpt_hip.h, rvpt_amd/native.py, rvpt_kernels.h)."""
    from rvpt_amd import native
    text = (ROOT / "include" / "rvpt_hip.h").read_text()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define (RVPT_HIP_[A-Z0-9_]+) (0x[0-9A-Fa-f]+|\d+)u?\b", text)}
    assert defs["RVPT_HIP_ABI_VERSION"] == native.ABI_VERSION and defs["RVPT_HIP_TILE"] == native.TILE and defs["RVPT_HIP_TILE_SHIFT"] == native.TILE_SHIFT
    for name in ("TRAVERSAL_BVH", "TRAVERSAL_BVH_ORDERED", "COUNT_SEGMENTS", "KERNEL_SIMPLE", "TIMING", "ACCUM_UNORM8", "BRUTE_MIXED_PACKETS", "BVH_PER_LANE"):
        assert defs["RVPT_HIP_" + name] == getattr(native, name), name
    known = 0
    for name, v in defs.items():
        if name in ("RVPT_HIP_TRAVERSAL_MASK", "RVPT_HIP_COUNT_SEGMENTS", "RVPT_HIP_KERNEL_SIMPLE", "RVPT_HIP_TIMING", "RVPT_HIP_ACCUM_UNORM8",
                    "RVPT_HIP_BRUTE_MIXED_PACKETS", "RVPT_HIP_BVH_PER_LANE"):
            known |= v
    assert defs["RVPT_HIP_FLAGS_KNOWN"] == known
    kh = (ROOT / "rvpt_amd" / "csrc" / "rvpt_kernels.h").read_text()
    assert int(re.search(r"constexpr uint32_t kTileShift = (\d+);", kh).group(1)) == native.TILE_SHIFT


def test_tile_slots_are_a_bijection_and_spread_columns_over_ranks():
    """The ownership rule (RVPT_HIP_TILE_SHIFT): slots number the tiles of any grid exactly once; tile() / untile() are inverse; and where the
    grid width is a multiple of the world size (1920 px, 8 ranks) no rank owns whole tile columns any more."""
    from rvpt_amd.distributed import tile_numpy, tile_slot, untile_numpy
    for tiles_x, tiles_y in ((120, 68), (7, 5), (1, 9), (9, 1), (8, 8), (3, 64)):
        ty, tx = np.meshgrid(np.arange(tiles_y), np.arange(tiles_x), indexing="ij")
        s = tile_slot(tx, ty, tiles_x)
        assert sorted(s.ravel().tolist()) == list(range(tiles_x * tiles_y))
    ty, tx = np.meshgrid(np.arange(68), np.arange(120), indexing="ij")
    rank = tile_slot(tx, ty, 120) % 8
    for r in range(8):
        cols = np.unique(tx[rank == r])
        assert cols.size == 120  # every rank touches every tile column (plain row-major numbering: 15 columns each)
        assert abs(int((rank == r).sum()) - 120 * 68 // 8) <= 1
    rng = np.random.RandomState(3)
    W, H = 100, 52  # partial edge tiles
    img = rng.rand(H, W, 4).astype(np.float32)
    for world in (1, 2, 3, 8):
        slots = [tile_numpy(img, r, world) for r in range(world)]
        n = max(x.shape[0] for x in slots)
        padded = np.stack([np.concatenate([x, np.zeros((n - x.shape[0], 4), np.float32)]) for x in slots])
        assert np.array_equal(untile_numpy(padded, W, H), img)


def test_null_and_invalid_arguments_are_errors_not_crashes(lib):
    from rvpt_amd import native
    assert lib.rvpt_hip_dispatch(None) == native.ERR_INVALID
    assert lib.rvpt_hip_wait(None) == native.ERR_INVALID
    assert lib.rvpt_hip_read(None, 0, None, 0) == native.ERR_INVALID
    assert lib.rvpt_hip_set_frame(None, None, None) == native.ERR_INVALID
    h = ctypes.c_void_p(None)
    assert lib.rvpt_hip_create(ctypes.byref(h), 0, 0, 16, 0, 1, 0) == native.ERR_INVALID  # zero width
    assert lib.rvpt_hip_create(ctypes.byref(h), 0, 16, 16, 2, 2, 0) == native.ERR_INVALID  # rank >= world
    assert b"bad geometry" in lib.rvpt_hip_last_error(None)
    assert lib.rvpt_bvh_build(None, 0, None, None, None) == native.ERR_INVALID
    lib.rvpt_hip_destroy(None)  # no-op


def test_create_without_gpu_reports_no_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rvpt_amd import native
    h = ctypes.c_void_p(None)
    assert lib.rvpt_hip_create(ctypes.byref(h), 0, 64, 64, 0, 1, 0) == native.ERR_NO_DEVICE
    with pytest.raises(native.NativeError):
        native.Context(64, 64)


def check_bvh(tris, nodes, idx):
    from rvpt_amd import native
    n = tris.shape[0]
    nd = nodes.view(native.NODE_DTYPE).reshape(-1)
    assert sorted(idx.tolist()) == list(range(n))  # a permutation
    st = tris[idx]
    seen = np.zeros(n, dtype=np.int32)
    max_depth = 0
    stack = [(0, 1)]
    while stack:
        i, depth = stack.pop()
        max_depth = max(max_depth, depth)
        b = nd[i]["bounds"]
        if nd[i]["count"] > 0:
            f, c = int(nd[i]["first"]), int(nd[i]["count"])
            seen[f:f + c] += 1
            p = st[f:f + c][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3)
            assert (p[:, 0] >= b[0]).all() and (p[:, 0] <= b[1]).all()
            assert (p[:, 1] >= b[2]).all() and (p[:, 1] <= b[3]).all()
            assert (p[:, 2] >= b[4]).all() and (p[:, 2] <= b[5]).all()
        else:
            l = int(nd[i]["first"])
            assert l + 1 < len(nd)
            for ch in (l, l + 1):  # children inside the parent
                cb = nd[ch]["bounds"]
                assert cb[0] >= b[0] and cb[1] <= b[1] and cb[2] >= b[2] and cb[3] <= b[3] and cb[4] >= b[4] and cb[5] <= b[5]
                stack.append((ch, depth + 1))
    assert (seen == 1).all()  # every primitive in exactly one leaf
    assert max_depth <= 62    # fits the traversal's 64-entry stack (intersection.glsl:363)
    return max_depth


def test_bvh_builder_default_scene(lib):
    from rvpt_amd import native, scene
    tris, _ = scene.default_scene()
    nodes, idx = native.build_bvh(tris)
    assert nodes.shape[0] <= 2 * 143 - 1
    check_bvh(tris, nodes, idx)


def test_bvh_builder_edge_cases(lib):
    from rvpt_amd import native, scene
    one = scene.make_triangles(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32), 0)
    nodes, idx = native.build_bvh(one)
    assert nodes.shape[0] == 1 and idx.tolist() == [0]
    # 300 identical triangles: centroid bounds are a point -> median splits must still terminate
    same = np.repeat(one, 300, axis=0)
    nodes, idx = native.build_bvh(same)
    check_bvh(same, nodes, idx)
    # a long sliver-sorted strip (degenerate for SAH)
    xs = np.arange(2000, dtype=np.float32)
    strip = scene.make_triangles(np.stack([np.stack([xs, 0 * xs, 0 * xs], 1), np.stack([xs + 1, 0 * xs, 0 * xs], 1),
                                           np.stack([xs, 0 * xs + 1, 0 * xs], 1)], 1), 0)
    nodes, idx = native.build_bvh(strip)
    d = check_bvh(strip, nodes, idx)
    assert d <= 40


def test_bvh_builder_heightfield_10k(lib):
    from rvpt_amd import native, scene
    tris, _ = scene.heightfield_scene(cells=70)
    nodes, idx = native.build_bvh(tris)
    check_bvh(tris, nodes, idx)
