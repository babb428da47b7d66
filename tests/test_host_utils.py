"""CPU tests of host-side helpers: OBJ ingestion (main.cpp:12-62 semantics), scene generators, image writers,
camera block, RenderSettings packing."""
import math
import struct
import zlib

import numpy as np
import pytest


def test_obj_reader_fan_triangulates_and_ignores_normals(tmp_path):
    from rvpt_amd import scene
    p = tmp_path / "m.obj"
    p.write_text("# c\no m\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4//1 -3//1 -2//1\nf 1 2 3\n")
    t = scene.load_obj_positions(p)
    assert t.shape == (4, 3, 3)
    assert np.array_equal(t[0], [[0, 0, 0], [1, 0, 0], [1, 1, 0]]) and np.array_equal(t[1], [[0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(t[2], t[0]) and np.array_equal(t[3], t[0])
    tri = scene.make_triangles(t, 1)
    assert tri.shape == (4, 16) and (tri[:, 12] == 1.0).all()
    assert np.allclose(tri[0, [3, 7, 11]], [0, 0, 1])  # face normal rides in the .w lanes (geometry.h:81-91)


def test_obj_roundtrip_of_generated_scene(tmp_path):
    from rvpt_amd import scene
    pos = scene.default_model_positions()
    scene.write_obj(tmp_path / "a.obj", pos)
    back = scene.load_obj_positions(tmp_path / "a.obj")
    assert np.array_equal(back, pos)


def test_default_scene_matches_main_cpp():
    from rvpt_amd import scene
    tris, mats = scene.default_scene()
    assert tris.shape == (143, 16) and mats.shape == (2, 12)
    assert (tris[:, 12] == 1.0).all()                              # load_model(..., 1)
    assert mats[0].tolist() == [1, 1, 1, 0, np.float32(0.1), np.float32(0.4), np.float32(0.6), 0, 0, 0, 0, 0]
    assert mats[1].tolist() == [1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    lo, hi = tris[:, [0, 1, 2]].min(0), tris[:, [0, 1, 2]].max(0)
    assert lo[1] > 0.02 and hi[1] < 1.64                           # SURVEY row 27 bbox


def test_synthetic_scene_sizes():
    from rvpt_amd import scene
    assert scene.cornell_scene()[0].shape[0] == 12 + 143 * 64      # BASELINE config 3: ~9.2k triangles
    assert scene.heightfield_scene(cells=20)[0].shape[0] == 800
    t, m = scene.materials_showcase_scene()
    assert set(t[:, 12].astype(int)) == {0, 1, 2, 3} and m.shape[0] == 4


def test_camera_block_and_matrix_convention():
    from rvpt_amd import Camera
    c = Camera(2.0)
    d = c.get_data()
    assert d.shape == (20,) and np.array_equal(d[:16].reshape(4, 4), np.eye(4, dtype=np.float32))
    assert d[16] == 2.0 and abs(d[17] - math.pi / 2) < 1e-7 and d[18] == 4.0 and d[19] == 0.0
    c.translate((0.0, 0.0, 1.0))
    assert np.allclose(c.get_data()[12:15], [0, 0, 1])
    c.rotate((90.0, 0.0, 0.0))  # about UP: forward (+Z) turns towards +X
    m = c.get_data()[:16].reshape(4, 4).T
    assert np.allclose(m[:3, 2], [1, 0, 0], atol=1e-6)
    assert c.get_data() is c.get_data()  # cached until a parameter changes
    c.set_fov(60.0)
    assert abs(c.get_data()[17] - math.radians(60)) < 1e-7


@pytest.mark.parametrize("pose", [(1, 2, 3, 30, -20, 10), (-0.5, 0.25, 4, -135, 60, -75), (0, 0, 0, 90, 90, 0), (2, -1, 0.5, 0, 0, 45)])
def test_camera_matrix_against_the_hand_expanded_product(pose):
    """construct_camera_matrix (camera.cpp:17-25): glm::translate, then glm::rotate about UP by rot.x, about RIGHT by rot.y,
    about FORWARD by rot.z (degrees, right-handed, post-multiplied, column-major storage).  Independent derivation: the
    product Ry(a)*Rx(b)*Rz(c) expanded by hand — the matrix glm itself names eulerAngleYXZ(a, b, c)."""
    from rvpt_amd import Camera
    tx, ty, tz, a, b, c = pose
    k = Camera(1.5)
    k.translation = np.array([tx, ty, tz], dtype=np.float64)
    k.rotation = np.array([a, b, c], dtype=np.float64)
    ca, sa, cb, sb, cc, sc = (f(math.radians(x)) for x in (a, b, c) for f in (math.cos, math.sin))
    want = np.array([[ca * cc + sa * sb * sc, -ca * sc + sa * sb * cc, sa * cb, tx],
                     [cb * sc, cb * cc, -sb, ty],
                     [-sa * cc + ca * sb * sc, sa * sc + ca * sb * cc, ca * cb, tz],
                     [0, 0, 0, 1]])
    got = k.get_data()[:16].reshape(4, 4).T.astype(np.float64)  # stored column-major
    assert np.allclose(got, want, atol=2e-7), (got, want)
    if pose == (0, 0, 0, 90, 90, 0):  # two quarter turns worked by hand: forward -> -Y, up -> +X, right -> -Z
        assert np.allclose(got[:3, 2], [0, -1, 0], atol=1e-7) and np.allclose(got[:3, 1], [1, 0, 0], atol=1e-7) and np.allclose(got[:3, 0], [0, 0, -1], atol=1e-7)


def test_render_settings_block_layout():
    from rvpt_amd import RenderSettings
    s = RenderSettings(max_bounces=5, aa=3, current_frame=7, camera_mode=0, split_ratio=(0.25, 0.75))
    b = s.pack()
    assert b.nbytes == 40
    assert struct.unpack("<iiIiiiiiff", b.tobytes()) == (5, 3, 7, 0, 9, 9, 9, 9, 0.25, 0.75)
    assert RenderSettings().current_frame == 1  # rvpt.h:81


def test_png_and_pfm_writers(tmp_path):
    from rvpt_amd import imageio
    rng = np.random.RandomState(0)
    img = rng.rand(5, 7, 4).astype(np.float32)
    imageio.write_pfm(tmp_path / "a.pfm", img)
    assert np.array_equal(imageio.read_pfm(tmp_path / "a.pfm"), img[..., :3])
    u8 = (img * 255).astype(np.uint8)
    imageio.write_png(tmp_path / "a.png", u8)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    ihdr = struct.unpack(">IIBBBBB", raw[16:29])
    assert ihdr[:4] == (7, 5, 8, 2)
    idat = raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8]
    rows = zlib.decompress(idat)
    assert len(rows) == 5 * (1 + 7 * 3) and rows[1:22] == u8[0, :, :3].tobytes()


class _FakeContext:
    """Stands in for native.Context so the host-side frame-counter logic can be tested without a GPU."""

    def __init__(self, *a, **k):
        self.frames = []

    def upload_scene(self, nodes, tris, mats):
        self.scene = (nodes, tris.copy(), mats.copy())

    def set_frame_fast(self, rs, camera):
        self.frames.append((rs.current_frame, rs.aa, rs.max_bounces, bytes(camera.tobytes())))

    def dispatch(self):
        pass

    def close(self):
        pass


def test_rvpt_mirror_frame_counter_follows_the_reference_reset_rule(monkeypatch):
    """RVPT::update (rvpt.cpp:96-111): first frame 0; +1 while nothing in PreviousFrameState changes; reset on camera,
    camera mode, split ratio or render-mode changes; NOT on aa / max_bounces (rvpt.cpp:21-29)."""
    from rvpt_amd import RVPT, native, scene
    monkeypatch.setattr(native, "Context", _FakeContext)
    r = RVPT(64, 32, traversal="bvh")
    tris, mats = scene.default_scene()
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    assert r.render_settings.current_frame == 1  # rvpt.h:81 default, overwritten by the first update()
    assert r.initialize()
    # initialize() built the BVH and uploaded triangles in leaf order (rvpt.cpp:83-86)
    nodes, up_tris, up_mats = r.context.scene
    assert nodes is not None and np.array_equal(up_tris, tris[r.primitive_indices]) and up_mats.shape == (2, 12)

    def step():
        r.update()
        r.draw()
        return r.render_settings.current_frame

    assert [step(), step(), step()] == [0, 1, 2]
    r.render_settings.aa = 4
    r.render_settings.max_bounces = 3
    assert step() == 3
    r.render_settings.top_right_render_mode = 5
    assert step() == 0 and step() == 1
    r.render_settings.split_ratio = (0.25, 0.5)
    assert step() == 0
    r.render_settings.split_ratio = [0.25, 0.5]      # same values in another container: keeps accumulating
    assert step() == 1
    r.render_settings.split_ratio[0] = 0.3           # mutated IN PLACE: compared by value like PreviousFrameState (rvpt.cpp:21-29)
    assert step() == 0 and step() == 1
    r.scene_camera.rotate((1.0, 0.0, 0.0))
    assert step() == 0 and step() == 1
    r.scene_camera.set_camera_mode(1)
    assert step() == 0
    r.scene_camera.set_fov(60.0)
    assert step() == 0 and step() == 1
    # what reached the boundary: (current_frame, aa, bounces, camera bytes)
    assert r.context.frames[3][:3] == (3, 4, 3)
    assert r.context.frames[0][3] != r.context.frames[-1][3]


def test_brute_force_context_uploads_no_nodes(monkeypatch):
    from rvpt_amd import RVPT, native, scene
    monkeypatch.setattr(native, "Context", _FakeContext)
    r = RVPT(32, 32, traversal="brute")
    tris, mats = scene.default_scene()
    r.add_triangle(tris[0])
    r.add_triangles(tris[1:])
    r.add_material(mats[0]); r.add_material(mats[1])
    r.initialize()
    assert r.context.scene[0] is None and r.context.scene[1].shape == (143, 16)


def test_bench_cli_defaults_and_cpu_baseline_leg(oracle, default_scene):
    """bench.py: argument defaults of the driver contract (N=1, finite K/W) and the cpu_baseline leg (oracle timed on
    a bounded sample) on a tiny frame."""
    import importlib.util
    import sys
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_module", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(bench)
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1 and 0 < a.steps <= 1000 and 0 <= a.warmup <= 100 and (a.width, a.height, a.aa, a.bounces) == (1920, 1080, 1, 8)
    assert a.traversal == "brute" and a.scene == "default"
    tris, mats, nodes = default_scene
    a.width, a.height = 64, 32
    from _util import identity_camera
    out = bench.cpu_baseline(a, tris, mats, nodes, identity_camera(2.0), 0.3)
    assert out["kind"] == "port" and out["unit"] == "Msamples/s" and out["value"] > 0 and out["cores"] >= 1 and "64x32" in out["sample"]


def test_bench_gpus_n_without_devices_fails_with_the_reason():
    """`python bench.py --gpus 8` without a launcher and without eight visible devices (none at all in the authoring container, one
    on a test box) exits non-zero with the reason and prints no JSON line — it never runs one GPU and labels the result 8."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("eight devices are visible here: the fan-out itself is covered by the -m gpu test")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RVPT_BENCH_SHARED_GPU")}
    res = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "8 GPUs requested" in res.stderr and "visible" in res.stderr
    assert "n_gpus" not in res.stdout


def _walk_binary_leaves(nodes, i=0):
    """Leaves of a reference-layout tree in the reference's visiting order (left first, intersection.glsl:404-407)."""
    out, stack = [], [0]
    while stack:
        i = stack.pop()
        if nodes["count"][i] > 0:
            out.append((int(nodes["first"][i]), int(nodes["count"][i])))
        else:
            f = int(nodes["first"][i])
            stack += [f + 1, f]
    return out


@pytest.mark.parametrize("loosen", [False, True])
def test_wide_form_keeps_the_reference_order_and_regroups_only_across_containing_boxes(loosen):
    """rvpt_bvh_wide_form (what upload_scene builds for rvpt_bvh4.hip; no GPU needed): (a) a depth-first, slot-order walk of the wide tree visits exactly
    the binary tree's leaves in the binary tree's left-first order; (b) every child slot carries the bounds of the binary node it stands for;
    (c) a binary inner node is collapsed away only if its box contains both children's boxes — with loosened boxes the violating nodes keep their own
    slot; (d) stack_need bounds what the walk stacks; (e) trees without a wide form say so."""
    from rvpt_amd import native, scene
    tris, _ = scene.cornell_scene()
    nodes_u32, idx = native.build_bvh(tris)
    nodes = np.ascontiguousarray(nodes_u32).view(native.NODE_DTYPE).reshape(-1).copy()
    if loosen:
        rng = np.random.RandomState(9)
        inner = np.flatnonzero(nodes["count"] == 0)
        pick = inner[rng.rand(inner.size) < 0.3]
        b = nodes["bounds"][pick].astype(np.float64)
        c, h = (b[:, 0::2] + b[:, 1::2]) / 2, (b[:, 1::2] - b[:, 0::2]) / 2 * rng.uniform(0.55, 1.3, (pick.size, 3))
        nb = np.empty_like(b)
        nb[:, 0::2], nb[:, 1::2] = c - h, c + h
        nodes["bounds"][pick] = nb.astype(np.float32)
    shift = 1
    while (1 << shift) <= max(len(nodes), tris.shape[0]):
        shift += 1
    wide, need = native.wide_form(nodes, shift)
    assert 0 < wide.shape[0] < len(nodes) and need >= 1
    heads = wide[:, 6, :].view(np.uint32)
    # binary nodes by their (bounds, first, count) so that a wide child can be traced back
    def key(i):
        return nodes["bounds"][i].tobytes()
    contains = lambda a, b: all(nodes["bounds"][b][2 * ax] >= nodes["bounds"][a][2 * ax] and nodes["bounds"][b][2 * ax + 1] <= nodes["bounds"][a][2 * ax + 1] for ax in range(3))
    # (a) + (d): walk the wide tree
    leaves, stack, deepest = [], [(0, None)], 0
    # stack of wide-node indices / leaf heads to visit; mirror of the kernel's push order (last child pushed first)
    todo = [("node", 0)]
    while todo:
        deepest = max(deepest, len(todo) - 1)
        kind, v = todo.pop()
        if kind == "leaf":
            leaves.append(v)
            continue
        kids = []
        for k in range(4):
            hd = int(heads[v, k])
            if hd == 0xFFFFFFFF:
                continue
            cnt, first = hd >> shift, hd & ((1 << shift) - 1)
            kids.append(("leaf", (first, cnt)) if cnt else ("node", first))
        assert len(kids) >= 2
        todo += kids[::-1]
    assert leaves == _walk_binary_leaves(nodes)
    assert deepest <= need
    # (b) + (c): rebuild which binary nodes each wide node lists, by descending the binary tree guided by the child count
    import collections
    queue = collections.deque([0])
    w = 0
    collapsed = set()
    while queue:
        b = queue.popleft()
        f = int(nodes["first"][b])
        want = [int(x) for x in heads[w] if x != 0xFFFFFFFF]
        listed = [f, f + 1]
        # expand until the listed nodes reproduce the wide node's children (bounds must match slot by slot)
        def matches(lst):
            if len(lst) != len(want):
                return False
            for k, n in enumerate(lst):
                got_bounds = np.array([wide[w, q, k] for q in range(6)], dtype=np.float32)
                if got_bounds.tobytes() != nodes["bounds"][n].tobytes():
                    return False
            return True
        frontier = [listed]
        found = None
        while frontier and found is None:
            nxt = []
            for lst in frontier:
                if matches(lst):
                    found = lst
                    break
                if len(lst) < len(want):
                    for pos, n in enumerate(lst):
                        if nodes["count"][n] == 0:
                            g = int(nodes["first"][n])
                            nxt.append(lst[:pos] + [g, g + 1] + lst[pos + 1:])
            frontier = nxt
        assert found is not None, f"wide node {w}: children are not a regrouping of binary node {b}'s descendants in order"
        # which binary nodes were collapsed away between b and the listed children
        def ancestors_between(lst):
            gone, cur = set(), [f, f + 1]
            while sorted(cur) != sorted(lst):
                for pos, n in enumerate(cur):
                    if n not in lst and nodes["count"][n] == 0:
                        g = int(nodes["first"][n])
                        gone.add(n)
                        cur = cur[:pos] + [g, g + 1] + cur[pos + 1:]
                        break
                else:
                    break
            return gone
        for n in ancestors_between(found):
            g = int(nodes["first"][n])
            assert contains(n, g) and contains(n, g + 1), f"binary node {n} was collapsed although its box does not contain its children"
            collapsed.add(n)
        for n in found:
            if nodes["count"][n] == 0:
                queue.append(n)
        w += 1
    assert w == wide.shape[0]
    assert (len(collapsed) > 0) and (not loosen or any(not (contains(n, int(nodes["first"][n])) and contains(n, int(nodes["first"][n]) + 1))
                                                       for n in np.flatnonzero(nodes["count"] == 0)))
    # (e) no wide form: a single-leaf tree, heads that do not pack
    single = np.zeros(1, dtype=native.NODE_DTYPE)
    single["count"] = 3
    assert native.wide_form(single, shift)[0].shape[0] == 0
    assert native.wide_form(nodes, 0)[0].shape[0] == 0


def test_obj_mtl_scene_description(tmp_path):
    """OBJ + MTL: usemtl / mtllib, the illum -> Material::Type mapping, the default material."""
    from rvpt_amd import scene
    (tmp_path / "s.mtl").write_text(
        "newmtl wall\nKd 0.7 0.6 0.5\nillum 2\n"
        "newmtl lamp\nKd 0 0 0\nKe 4 5 6\n"
        "newmtl chrome\nKd 0.1 0.1 0.1\nKs 0.9 0.8 0.7\nillum 3\n"
        "newmtl glass\nKd 1 1 1\nNi 1.33\nillum 7\n"
        "newmtl faded\nKd 0.2 0.3 0.4\nd 0.5\n")
    (tmp_path / "s.obj").write_text(
        "mtllib s.mtl\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\n"
        "f 1 2 3\n"                      # before any usemtl -> "default"
        "usemtl glass\nf 1 2 3 4\n"     # quad -> 2 triangles
        "usemtl wall\nf 1 3 4\nusemtl nosuch\nf 2 3 4\n"
        "usemtl chrome\nf 1 2 4\nusemtl lamp\nf -4 -3 -2\nusemtl faded\nf 1 2 3\nusemtl glass\nf 4 3 2\n")
    tris, mats, names = scene.load_obj_scene(tmp_path / "s.obj")
    assert names == ["default", "glass", "wall", "chrome", "lamp", "faded"]
    assert tris[:, 12].tolist() == [0, 1, 1, 2, 0, 3, 4, 5, 1]
    f32 = lambda *v: np.asarray(v, dtype=np.float32)
    assert np.array_equal(mats[0][:9], f32(1, 1, 1, 0, 0, 0, 0, 0, scene.LAMBERT))
    assert np.array_equal(mats[1][:9], f32(1, 1, 1, 1.33, 0, 0, 0, 0, scene.DIELECTRIC))
    assert np.array_equal(mats[2][:9], f32(0.7, 0.6, 0.5, 0, 0, 0, 0, 0, scene.LAMBERT))
    assert np.array_equal(mats[3][:9], f32(0.9, 0.8, 0.7, 0, 0, 0, 0, 0, scene.MIRROR))
    assert np.array_equal(mats[4][:9], f32(0, 0, 0, 0, 4, 5, 6, 0, scene.LAMBERT))
    assert np.array_equal(mats[5][:9], f32(0.2, 0.3, 0.4, 1.5, 0, 0, 0, 0, scene.DIELECTRIC))


def test_obj_mtl_roundtrip_of_the_showcase_scene(tmp_path):
    from rvpt_amd import scene
    tris, mats = scene.materials_showcase_scene()
    scene.write_obj_scene(tmp_path / "show.obj", tris, mats)
    t2, m2, names = scene.load_obj_scene(tmp_path / "show.obj")
    assert np.array_equal(t2[:, :12], tris[:, :12])  # positions + packed face normals
    assert np.array_equal(m2[t2[:, 12].astype(int)], mats[tris[:, 12].astype(int)])  # same material per triangle


def test_bvh_build_is_identical_for_any_thread_count(monkeypatch):
    """rvpt_bvh_build splits large scenes over host threads; node numbering and leaf order must not depend on it."""
    from rvpt_amd import native, scene
    tris, _ = scene.heightfield_scene(cells=150)  # 45 000 triangles: above the multi-thread threshold
    out = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RVPT_BVH_THREADS", threads)
        nodes, idx = native.build_bvh(tris)
        out.append((nodes.tobytes(), idx.tobytes()))
    assert out[0] == out[1] == out[2]
    assert sorted(np.frombuffer(out[0][1], dtype=np.uint32).tolist()) == list(range(tris.shape[0]))


def test_launch_sizes_are_as_few_and_as_equal_as_the_batch_allows():
    """renderer.launch_sizes (== rvpt_host.cpp::launch_sizes, checked by host_selftest): the driver's 20-step run at 8 ranks (batch 64)
    is ONE launch (measured best on a small tile share, profiles/r03_launch_shapes.txt); at batch 8 it is 7 + 7 + 6."""
    from rvpt_amd.renderer import launch_sizes
    assert launch_sizes(20, 64) == [20]
    assert launch_sizes(20, 8) == [7, 7, 6]
    assert launch_sizes(200, 8) == [8] * 25
    assert launch_sizes(5, 64) == [5]
    assert launch_sizes(16, 1) == [1] * 16
    assert launch_sizes(0, 8) == []
    assert launch_sizes(20, 64, 6) == [20]
    assert launch_sizes(65, 64) == [33, 32]
    for frames in range(1, 200):
        for batch in (1, 3, 8, 64):
            sizes = launch_sizes(frames, batch)
            assert sum(sizes) == frames and max(sizes) <= batch and max(sizes) - min(sizes) <= 1
            assert len(sizes) == -(-frames // batch)


def test_fast_division_is_the_integer_quotient():
    """rvpt_kernels.h: FastDiv — the multiply-high division by tiles_x / work items per frame in the frame kernels' work-index decode — against numpy's //, for
    divisors of every kind (1, powers of two, their neighbours, primes, the tile counts and frame sizes of the bench configurations, 2^31 + 1, 2^32 - 1) and
    numerators at every boundary: multiples of the divisor, one less, one more, the top of the 32-bit range, random."""
    from rvpt_amd import native
    rng = np.random.default_rng(11)
    divisors = [1, 2, 3, 5, 7, 13, 16, 17, 31, 32, 33, 120, 127, 240, 255, 256, 257, 1020, 8160 * 256, 32640 * 256, 65535, 65536, 65537, 2088960, 8355840,
                (1 << 31) - 1, 1 << 31, (1 << 31) + 1, (1 << 32) - 1] + [int(d) for d in rng.integers(1, 1 << 32, 200, dtype=np.uint64)]
    for d in divisors:
        k = rng.integers(0, (1 << 32) // d + 1, 400, dtype=np.uint64) * d
        x = np.concatenate([k, k + d - 1, k - 1, k + 1, rng.integers(0, 1 << 32, 2000, dtype=np.uint64),
                            np.array([0, 1, d - 1, d, d + 1, (1 << 32) - 1, (1 << 32) - 2, 1 << 31], dtype=np.uint64)])
        x = (x % (1 << 32)).astype(np.uint32)
        assert np.array_equal(native.fast_div(x, d), (x.astype(np.uint64) // d).astype(np.uint32)), d


def test_claim_order_is_a_bijection_that_spreads_every_claim_over_the_frame():
    """rvpt_claim_order (rvpt_kernels.h: claim_order_block, rvpt_abi.hip: plan_claim_order; no GPU needed): the order in which one-frame launches of the packet
    kernel deal a frame's 16 x 4 blocks is a permutation of the blocks for every image size, rank share and group size (a block dealt twice or never would be a
    wrong image), keeps groups of g consecutive blocks together, never leaves 32 bits, and any eight consecutive groups of the order — one 512-item claim at
    g = 1 — land in at least four, on average more than six, different eighths of the frame (what the order is for: a claim holds its share of sky and of model); frames beyond
    ~17 M pixels keep the tile-linear order."""
    from rvpt_amd import native
    sizes = [(1920, 1080, 1), (1920, 1080, 8), (1920, 1080, 3), (3840, 2160, 1), (7680, 4320, 1), (256, 256, 1), (1000, 700, 1), (640, 360, 3), (16, 16, 1), (64, 64, 1), (5120, 2880, 1), (16384, 16384, 1)]
    for w, h, world in sizes:
        tiles = -(-w // 16) * -(-h // 16)
        n_work_frame = ((tiles + world - 1) // world) * 256  # rank 0's share
        for g in (1, 2, 4, 8):
            order, (groups, stride, shift) = native.claim_order(n_work_frame, g)
            blocks = n_work_frame // 64
            assert np.array_equal(np.sort(order), np.arange(blocks, dtype=np.uint32)), (w, h, world, g)
            if groups == 0:
                assert np.array_equal(order, np.arange(blocks, dtype=np.uint32))  # too small (or not whole groups): the tile-linear order
                assert blocks // g < 16 or n_work_frame % (64 * g) or (blocks // g) ** 2 / 16.62 > 1 << 32  # (... or the products would leave 32 bits)
                continue
            assert groups == blocks // g and (1 << shift) == g and np.gcd(stride, groups) == 1 and (groups - 1) * stride < 1 << 32
            assert np.array_equal(order % g, np.arange(blocks, dtype=np.uint32) % g)  # a group stays together, in order
            run = (order[::g] // g).astype(np.int64)  # the groups in claim order
            if groups >= 64:
                eighth = run * 8 // groups
                windows = np.sort(np.lib.stride_tricks.sliding_window_view(eighth, 8)[:: max(1, groups // 4096)], axis=1)
                distinct = 1 + (np.diff(windows, axis=1) != 0).sum(axis=1)
                assert distinct.min() >= 4 and (distinct.mean() >= 6.0 or stride < 0.2 * groups), (w, h, world, g, distinct.min(), distinct.mean())  # (frames beyond 3840 x 2160: a shorter stride)
    assert native.claim_order(2088960, 3)[1][0] == 0 and native.claim_order(2088960, 0)[1][0] == 0  # group sizes the library does not use: identity


def test_quantised_wide_nodes_contain_the_exact_child_boxes_and_leaves_keep_theirs():
    """rvpt_bvh_quant_form (what upload_scene builds for trace_bvh4q, RVPT_HIP_BVH_QUANT=1; no GPU needed): every child box of the 64-byte form,
    origin + q * scale, CONTAINS the exact child box of the 128-byte form (the walk's inner boxes may only cull less, never more); scale is a power of
    two and origin + 255 * scale stays inside `extent`; the heads are the 128-byte form's; the leaf boxes are the binary leaves' own, by first triangle;
    a tree with a box that does not contain a child, or with two leaves on one triangle, has no quantised form."""
    from rvpt_amd import native, scene

    def far_flat_walls():  # axis-aligned quads far from the origin: wide nodes that are FLAT on an axis (two coplanar triangles), corners of magnitude 100
        quads = []
        for i in range(6):
            for j in range(6):
                x, z = 100.0 + i, -50.0 + j
                quads += [[x, 4.0, z], [x + 1, 4.0, z], [x, 4.0, z + 1], [x + 1, 4.0, z], [x + 1, 4.0, z + 1], [x, 4.0, z + 1]]
        return scene.make_triangles(np.array(quads, dtype=np.float32).reshape(-1, 3, 3), 0), None

    for make in (scene.cornell_scene, scene.default_scene, far_flat_walls):
        tris, _ = make()
        nodes_u32, idx = native.build_bvh(tris)
        nodes = np.ascontiguousarray(nodes_u32).view(native.NODE_DTYPE).reshape(-1).copy()
        shift = 1
        while (1 << shift) <= max(len(nodes), tris.shape[0]):
            shift += 1
        wide, _ = native.wide_form(nodes, shift)
        quant, boxes, extent = native.quant_form(nodes, shift, tris.shape[0])
        assert quant.shape[0] == wide.shape[0] > 0
        origin, scale = quant[:, 0:3].view(np.float32).astype(np.float64), quant[:, 3:6].view(np.float32).astype(np.float64)
        m, e = np.frexp(scale)
        assert np.all(m == 0.5) and np.all(scale > 0)  # powers of two
        heads = wide[:, 6, :].view(np.uint32)
        assert np.array_equal(quant[:, 12:16], heads)
        assert np.all(np.abs(origin) <= extent) and np.all(np.abs(origin + 255.0 * scale) <= extent)
        used = heads != 0xFFFFFFFF
        worst = 0.0
        for ax in range(3):
            for k in range(4):
                qmin = ((quant[:, 6 + 2 * ax] >> (8 * k)) & 0xFF).astype(np.float64)
                qmax = ((quant[:, 7 + 2 * ax] >> (8 * k)) & 0xFF).astype(np.float64)
                lo, hi = origin[:, ax] + qmin * scale[:, ax], origin[:, ax] + qmax * scale[:, ax]
                bmin, bmax = wide[:, 2 * ax, k].astype(np.float64), wide[:, 2 * ax + 1, k].astype(np.float64)
                u = used[:, k]
                assert np.all(lo[u] <= bmin[u]) and np.all(hi[u] >= bmax[u])
                assert np.all(bmin[u] - lo[u] <= scale[u, ax]) and np.all(hi[u] - bmax[u] <= scale[u, ax])  # ... and by less than one step
                assert np.all(qmin[~u] > qmax[~u])  # an unused slot: an empty interval
        leaves = np.flatnonzero(nodes["count"] > 0)
        assert np.array_equal(boxes[nodes["first"][leaves], :6], nodes["bounds"][leaves])
        assert extent >= np.abs(nodes["bounds"]).max()
        # no quantised form: a child that sticks out of its parent / two leaves on one triangle
        bad = nodes.copy()
        inner = np.flatnonzero(bad["count"] == 0)
        bad["bounds"][bad["first"][inner[3]]][1] += 1000.0
        assert native.quant_form(bad, shift, tris.shape[0])[0].shape[0] == 0
        bad = nodes.copy()
        bad["first"][leaves[1]] = bad["first"][leaves[0]]
        assert native.quant_form(bad, shift, tris.shape[0])[0].shape[0] == 0
