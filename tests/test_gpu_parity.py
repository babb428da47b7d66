"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the committed fixtures.

Bar (BASELINE.json north_star): relative per-image L2 <= 1e-4 on float radiance.  Both sides implement the
same arithmetic specification independently, so the expected distance is 0; the tests assert the 1e-4 bar
and additionally that (almost) every pixel is bit-identical, which is what makes the bar meaningful for a
chaotic integrator."""
import os
from pathlib import Path

import numpy as np
import pytest

from _util import GOLDEN, identity_camera, rel_l2, scene_by_name

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

TOL = 1e-4  # north_star: relative per-pixel L2 on identical scene/camera/RNG seed


@pytest.fixture(scope="module")
def native():
    from rvpt_amd import build, native as n
    build.build_native()
    build.build_native_debug()  # the laboratory build: selftests, opt-in walks, knobs (Context(lab=True) / RVPT_HIP_LAB=1)
    n.load()
    assert n.device_count() >= 1
    return n


def gpu_frames(native, scene, cam, W, H, traversal, frames, aa=1, max_bounces=8, flags=0, world=1, rank=0):
    tris, mats, nodes = scene
    fl = flags | {"bvh": native.TRAVERSAL_BVH, "brute": native.TRAVERSAL_BRUTE, "bvh_ordered": native.TRAVERSAL_BVH_ORDERED}[traversal]
    ctx = native.Context(W, H, 0, rank, world, fl)
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        out = []
        from rvpt_amd import RenderSettings
        for f in frames:
            rs = RenderSettings(max_bounces=max_bounces, aa=aa, current_frame=f)
            ctx.set_frame(rs.pack(), cam)
            ctx.dispatch()
            out.append(ctx.read())
        extra = ctx.stats() if (fl & native.COUNT_SEGMENTS) else None
        return out, extra
    finally:
        ctx.close()


def oracle_frames(oracle, scene, cam, W, H, traversal, frames, aa=1, max_bounces=8):
    tris, mats, nodes = scene
    trav = {"bvh": oracle.TRAVERSAL_BVH, "brute": oracle.TRAVERSAL_BRUTE, "bvh_ordered": oracle.TRAVERSAL_BVH_ORDERED}[traversal]
    out, prev, seg = [], None, 0
    for f in frames:
        s = oracle.settings_bytes(max_bounces=max_bounces, aa=aa, current_frame=f)
        img, stats = oracle.render(s, cam, nodes, tris, mats, W, H, trav, prev=prev)
        prev = img
        seg += int(stats[0])
        out.append(img)
    return out, seg


def assert_parity(got, ref, what, max_mismatch_frac=1e-4):
    d = rel_l2(got, ref)
    mism = int((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    n = got.shape[0] * got.shape[1]
    assert d <= TOL, f"{what}: rel L2 {d:.3e} > {TOL} ({mism}/{n} pixels differ)"
    assert mism <= max_mismatch_frac * n, f"{what}: {mism}/{n} pixels not bit-identical (rel L2 {d:.3e})"


@pytest.mark.parametrize("traversal", ["brute", "bvh"])
def test_default_scene_256_frame0(native, oracle, traversal):
    """BASELINE config 0 (default scene 256x256 1spp) — like-for-like per traversal."""
    sc = scene_by_name("default")
    cam = identity_camera(1.0)
    got, st = gpu_frames(native, sc, cam, 256, 256, traversal, [0], flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 256, 256, traversal, [0])
    assert_parity(got[0], ref[0], f"default 256^2 {traversal}")
    assert st == (seg, 256 * 256)  # same number of path segments traced


@pytest.mark.parametrize("case", sorted(p.stem for p in GOLDEN.glob("*.npz")))
def test_against_committed_fixtures(native, case):
    """Frames 0..3 (aa=2, temporal accumulation) against tests/golden — no oracle at run time."""
    fx = np.load(GOLDEN / f"{case}.npz")
    sc = scene_by_name(case.split("_")[0])
    trav = {"brute": "brute", "bvh": "bvh", "bvhordered": "bvh_ordered"}[case.split("_")[-1]]
    got, st = gpu_frames(native, sc, fx["camera"], int(fx["width"]), int(fx["height"]), trav, [0, 1, 2, 3], aa=int(fx["aa"]),
                         max_bounces=int(fx["max_bounces"]), flags=native.COUNT_SEGMENTS)
    assert_parity(got[0], fx["frame0"], f"{case} frame0", max_mismatch_frac=0)
    assert_parity(got[3], fx["frame3"], f"{case} frame3", max_mismatch_frac=0)


@pytest.mark.parametrize("traversal", ["brute", "bvh"])
def test_showcase_materials_temporal_aa(native, oracle, traversal):
    """mirror + dielectric + emitter, aa=3, frames 0..2 — exercises every branch of integrator_Kajiya."""
    from rvpt_amd import Camera
    sc = scene_by_name("showcase")
    c = Camera(160 / 96)
    c.translation = np.array([0.3, 1.1, -2.2])
    c.rotation = np.array([-8.0, 6.0, 0.0])
    cam = c.get_data()
    got, _ = gpu_frames(native, sc, cam, 160, 96, traversal, [0, 1, 2], aa=3)
    ref, _ = oracle_frames(oracle, sc, cam, 160, 96, traversal, [0, 1, 2], aa=3)
    for f in range(3):
        assert_parity(got[f], ref[f], f"showcase {traversal} frame {f}")


def test_streamed_brute_force_cornell_9k(native, oracle):
    """BASELINE config 2 geometry (Cornell + 9152-triangle model = 9164 triangles > LDS-resident limit):
    the chunk-streamed kernel against the oracle at a size the oracle finishes in seconds."""
    sc = scene_by_name("cornell")
    assert sc[0].shape[0] == 9164
    from rvpt_amd import Camera
    c = Camera(96 / 64)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()
    got, st = gpu_frames(native, sc, cam, 96, 64, "brute", [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 96, 64, "brute", [0, 1], aa=2)
    assert_parity(got[0], ref[0], "cornell brute frame0")
    assert_parity(got[1], ref[1], "cornell brute frame1")
    assert st[0] == seg
    gotb, _ = gpu_frames(native, sc, cam, 96, 64, "bvh", [0, 1], aa=2)
    refb, _ = oracle_frames(oracle, sc, cam, 96, 64, "bvh", [0, 1], aa=2)
    assert_parity(gotb[1], refb[1], "cornell bvh frame1")


def test_full_hd_default_scene_vs_oracle(native, oracle):
    """BASELINE config 1 at full size: 1920x1080, 1 spp, 8 bounces, default scene and camera."""
    sc = scene_by_name("default")
    cam = identity_camera(1920 / 1080)
    got, st = gpu_frames(native, sc, cam, 1920, 1080, "brute", [0], flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 1920, 1080, "brute", [0])
    assert_parity(got[0], ref[0], "1920x1080 brute")
    assert st == (seg, 1920 * 1080)
    # the reference's integer-division dispatch leaves rows 1072..1079 unwritten; we render them
    assert np.abs(got[0][1072:]).sum() > 0


def test_full_hd_partition_and_kernel_invariance(native):
    """Size-independent properties at 1920x1080: (a) regenerating and one-pixel-per-lane kernels produce the
    same bits, (b) any tile partition (2 and 3 ranks, emulated on one GPU) sums to the unsplit frame,
    (c) brute force equals BVH except for a handful of tie / slab-culling pixels."""
    sc = scene_by_name("default")
    cam = identity_camera(1920 / 1080)
    full, _ = gpu_frames(native, sc, cam, 1920, 1080, "brute", [0, 1])
    simple, _ = gpu_frames(native, sc, cam, 1920, 1080, "brute", [0, 1], flags=native.KERNEL_SIMPLE)
    assert np.array_equal(full[1], simple[1])
    for world in (2, 3):
        acc = np.zeros_like(full[1])
        owned = np.zeros(full[1].shape[:2], dtype=np.int32)
        for rank in range(world):
            part, _ = gpu_frames(native, sc, cam, 1920, 1080, "brute", [0, 1], world=world, rank=rank)
            acc += part[1]
            ty, tx = np.meshgrid(np.arange(1080) // 16, np.arange(1920) // 16, indexing="ij")
            from rvpt_amd.distributed import tile_slot
            mine = (tile_slot(tx, ty, 120) % world) == rank
            owned += mine
            assert (part[1][~mine] == 0).all()
        assert (owned == 1).all()
        assert np.array_equal(acc, full[1])
    bvh, _ = gpu_frames(native, sc, cam, 1920, 1080, "bvh", [0, 1])
    differing = int((bvh[1] != full[1]).any(axis=2).sum())
    assert differing <= 1e-3 * 1920 * 1080, differing


def test_edge_inputs(native, oracle):
    """Empty scene, 1-pixel-wide and non-multiple-of-16 images, zero bounces, one triangle."""
    cam = identity_camera(37 / 21)
    empty = (np.zeros((0, 16), np.float32), np.zeros((0, 12), np.float32), None)
    got, _ = gpu_frames(native, empty, cam, 37, 21, "brute", [0])
    ref, _ = oracle_frames(oracle, empty, cam, 37, 21, "brute", [0])
    assert np.array_equal(got[0], ref[0])
    sc = scene_by_name("default")
    for W, H in [(1, 1), (17, 1), (1, 33), (31, 47)]:
        cam = identity_camera(W / H)
        got, _ = gpu_frames(native, sc, cam, W, H, "brute", [0, 1], aa=2)
        ref, _ = oracle_frames(oracle, sc, cam, W, H, "brute", [0, 1], aa=2)
        assert np.array_equal(got[1], ref[1]), (W, H)
    cam = identity_camera(1.0)
    got, _ = gpu_frames(native, sc, cam, 32, 32, "bvh", [0], max_bounces=0)
    assert (got[0] == 0).all()  # integrators.glsl:574: the loop never runs -> vec3(0)
    got, _ = gpu_frames(native, sc, cam, 32, 32, "brute", [0], max_bounces=1)
    ref, _ = oracle_frames(oracle, sc, cam, 32, 32, "brute", [0], max_bounces=1)
    assert np.array_equal(got[0], ref[0])


def test_rgba8_read_matches_reference_image_format(native, oracle):
    sc = scene_by_name("default")
    cam = identity_camera(2.0)
    tris, mats, nodes = sc
    from rvpt_amd import RenderSettings
    ctx = native.Context(128, 64, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        ctx.upload_scene(nodes, tris, mats)
        ctx.set_frame(RenderSettings(current_frame=0).pack(), cam)
        ctx.dispatch()
        f32 = ctx.read(native.FORMAT_RGBA32F)
        u8 = ctx.read(native.FORMAT_RGBA8_UNORM)
    finally:
        ctx.close()
    assert np.array_equal(u8, oracle.quantize_rgba8(f32))


def test_accumulator_checkpoint_roundtrip(native, oracle):
    """write_accum + frame f continues an accumulation exactly (resume)."""
    sc = scene_by_name("default")
    cam = identity_camera(1.0)
    straight, _ = gpu_frames(native, sc, cam, 64, 64, "brute", [0, 1, 2, 3])
    tris, mats, nodes = sc
    from rvpt_amd import RenderSettings
    ctx = native.Context(64, 64, 0, 0, 1, 0)
    try:
        ctx.upload_scene(None, tris, mats)
        ctx.write_accum(straight[1])
        assert np.array_equal(ctx.read(), straight[1])
        for f in (2, 3):
            ctx.set_frame(RenderSettings(current_frame=f).pack(), cam)
            ctx.dispatch()
        assert np.array_equal(ctx.read(), straight[3])
    finally:
        ctx.close()


def test_error_behaviour(native):
    from rvpt_amd import RenderSettings
    sc = scene_by_name("default")
    tris, mats, nodes = sc
    cam = identity_camera(1.0)
    ctx = native.Context(32, 32, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        with pytest.raises(native.NativeError) as e:
            ctx.dispatch()
        assert e.value.code == native.ERR_INVALID
        with pytest.raises(native.NativeError) as e:
            ctx.upload_scene(None, tris, mats)  # BVH context without nodes
        assert e.value.code == native.ERR_INVALID
        bad = tris.copy()
        bad[5, 12] = 9.0
        with pytest.raises(native.NativeError) as e:
            ctx.upload_scene(nodes, bad, mats)
        assert e.value.code == native.ERR_INVALID and "material index" in str(e.value)
        ctx.upload_scene(nodes, tris, mats)
        ctx.set_frame(RenderSettings(top_right_render_mode=10, bottom_left_render_mode=-1, camera_mode=-1).pack(), cam)  # all valid
        with pytest.raises(native.NativeError) as e:
            ctx.set_frame(RenderSettings(aa=0).pack(), cam)
        assert e.value.code == native.ERR_INVALID
        import ctypes
        small = np.zeros(16, np.float32)
        rc = native.load().rvpt_hip_read(ctx._h, 0, small.ctypes.data_as(ctypes.c_void_p), small.nbytes)
        assert rc == native.ERR_SIZE
    finally:
        ctx.close()


def test_rvpt_host_interface_accumulates_and_resets(native, oracle):
    """The class-RVPT mirror: update()/draw() frame counter follows rvpt.cpp:102-111."""
    from rvpt_amd import RVPT, scene
    r = RVPT(96, 64, traversal="bvh")
    tris, mats = scene.default_scene()
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    assert r.initialize()
    try:
        frames = []
        for _ in range(3):
            r.update()
            frames.append(r.render_settings.current_frame)
            r.draw()
        assert frames == [0, 1, 2]
        r.render_settings.aa = 2  # aa is not part of the reset key (rvpt.cpp:21-29)
        r.update(); r.draw()
        assert r.render_settings.current_frame == 3
        r.scene_camera.translate((0.0, 0.0, -0.5))
        r.update(); r.draw()
        assert r.render_settings.current_frame == 0
        img = r.read_frame()
        s = oracle.settings_bytes(aa=2, current_frame=0)
        ref, _ = oracle.render(s, r.scene_camera.get_data(), r.bvh_nodes, r.sorted_triangles, mats, 96, 64, oracle.TRAVERSAL_BVH)
        assert np.array_equal(img, ref)
    finally:
        r.shutdown()


def test_gather_untile_on_gpu(native):
    """The multi-GPU read path on one GPU: two contexts render the two halves of the tile partition, their
    tile-linear device buffers are viewed zero-copy as torch tensors (what the RCCL gather sends), stacked
    like the gather output and un-tiled by the library; result equals the single-context frame."""
    import torch
    from rvpt_amd import RenderSettings
    from rvpt_amd.distributed import _DeviceBuffer
    sc = scene_by_name("default")
    tris, mats, nodes = sc
    W, H, world = 200, 120, 2
    cam = identity_camera(W / H)
    full, _ = gpu_frames(native, sc, cam, W, H, "brute", [0, 1], aa=2)
    ctxs = [native.Context(W, H, 0, r, world, 0) for r in range(world)]
    try:
        slots = []
        for c in ctxs:
            c.upload_scene(None, tris, mats)
            for f in (0, 1):
                c.set_frame(RenderSettings(aa=2, current_frame=f).pack(), cam)
                c.dispatch()
            c.wait()
            ptr, nbytes, slot_bytes = c.tile_buffer()
            assert nbytes <= slot_bytes
            slots.append(torch.as_tensor(_DeviceBuffer(ptr, slot_bytes // 4), device="cuda:0"))
        gathered = torch.stack(slots).contiguous()
        out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        ctxs[0].untile(gathered.data_ptr(), gathered.shape[1] * 4, world, out.data_ptr())
        assert np.array_equal(out.cpu().numpy(), full[1])
    finally:
        for c in ctxs:
            c.close()


def test_long_accumulation_chain_with_frames_in_flight(native, oracle):
    """16 frames dispatched back to back (no wait in between): the frame kernels overlap in flight and the blend
    passes must still apply the running mean in dispatch order — compare the final accumulator with the oracle."""
    sc = scene_by_name("default")
    tris, mats, nodes = sc
    W, H, n = 80, 48, 16
    cam = identity_camera(W / H)
    from rvpt_amd import RenderSettings
    for traversal, fl in (("brute", 0), ("bvh", native.TRAVERSAL_BVH)):
        ctx = native.Context(W, H, 0, 0, 1, fl)
        try:
            ctx.upload_scene(nodes if traversal == "bvh" else None, tris, mats)
            for f in range(n):
                ctx.set_frame(RenderSettings(aa=1, current_frame=f).pack(), cam)
                ctx.dispatch()
            assert ctx.launch_info()[3] >= 1
            got = ctx.read()
        finally:
            ctx.close()
        ref, _ = oracle_frames(oracle, sc, cam, W, H, traversal, list(range(n)))
        assert np.array_equal(got, ref[-1]), traversal


def test_reference_format_accumulation_unorm8(native, oracle):
    """RVPT_HIP_ACCUM_UNORM8: the running mean goes through the rgba8 temporal image every frame like upstream
    (compute_pass.comp:41-42,146-148,165).  Oracle chain: quantise each frame's output, feed the dequantised
    texels back as `prev`."""
    sc = scene_by_name("default")
    tris, mats, nodes = sc
    W, H = 96, 64
    cam = identity_camera(W / H)
    from rvpt_amd import RenderSettings
    ctx = native.Context(W, H, 0, 0, 1, native.ACCUM_UNORM8)
    try:
        ctx.upload_scene(None, tris, mats)
        prev, ref_u8 = None, None
        for f in range(5):
            ctx.set_frame(RenderSettings(aa=2, current_frame=f).pack(), cam)
            ctx.dispatch()
            s = oracle.settings_bytes(aa=2, current_frame=f)
            out, _ = oracle.render(s, cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BRUTE, prev=prev)
            ref_u8 = oracle.quantize_rgba8(out)
            prev = oracle.dequantize_rgba8(ref_u8)
        got_u8 = ctx.read(native.FORMAT_RGBA8_UNORM)
        got_f32 = ctx.read(native.FORMAT_RGBA32F)
    finally:
        ctx.close()
    assert np.array_equal(got_u8, ref_u8)          # north_star asks +-1 LSB; the two sides agree exactly
    assert np.array_equal(got_f32, prev)


def test_million_triangle_heightfield(native, oracle):
    """BASELINE config 3/4 geometry class: the 708x708 heightfield (1 002 528 triangles, 64 MB of records, 49 MB of
    BVH nodes).  BVH traversal and the LDS-streamed brute-force loop against the oracle at a size it finishes fast."""
    from rvpt_amd import Camera, scene
    tris, mats = scene.heightfield_scene()
    assert tris.shape[0] == 1002528
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    c = Camera(64 / 40)
    c.translation = np.array([0.0, 2.5, -5.0])
    c.rotation = np.array([0.0, 25.0, 0.0])
    cam = c.get_data()
    got, st = gpu_frames(native, sc, cam, 64, 40, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 64, 40, "bvh", [0, 1], aa=2)
    assert_parity(got[1], ref[1], "heightfield 1M bvh")
    assert st[0] == seg and seg > 64 * 40 * 2 * 2  # paths do bounce off the terrain
    got, _ = gpu_frames(native, sc, cam, 32, 20, "brute", [0])
    ref, _ = oracle_frames(oracle, sc, cam, 32, 20, "brute", [0])
    assert_parity(got[0], ref[0], "heightfield 1M brute (streamed)")


def _frames_with_settings(native, oracle, sc, cam, W, H, traversal, settings_kw, frames=(0, 1)):
    """GPU and oracle frames for arbitrary RenderSettings fields (modes, split, camera_mode, aa, bounces)."""
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    fl = native.COUNT_SEGMENTS | {"bvh": native.TRAVERSAL_BVH, "brute": 0, "bvh_ordered": native.TRAVERSAL_BVH_ORDERED}[traversal]
    ctx = native.Context(W, H, 0, 0, 1, fl)
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        for f in frames:
            ctx.set_frame(RenderSettings(current_frame=f, **settings_kw).pack(), cam)
            ctx.dispatch()
        got, st = ctx.read(), ctx.stats()
    finally:
        ctx.close()
    okw = dict(max_bounces=settings_kw.get("max_bounces", 8), aa=settings_kw.get("aa", 1), camera_mode=settings_kw.get("camera_mode", 0),
               modes=(settings_kw.get("top_left_render_mode", 9), settings_kw.get("top_right_render_mode", 9),
                      settings_kw.get("bottom_left_render_mode", 9), settings_kw.get("bottom_right_render_mode", 9)),
               split=settings_kw.get("split_ratio", (0.5, 0.5)))
    prev, seg = None, 0
    trav = {"bvh": oracle.TRAVERSAL_BVH, "brute": oracle.TRAVERSAL_BRUTE, "bvh_ordered": oracle.TRAVERSAL_BVH_ORDERED}[traversal]
    for f in frames:
        ref, stats = oracle.render(oracle.settings_bytes(current_frame=f, **okw), cam, nodes, tris, mats, W, H, trav, prev=prev)
        prev = ref
        seg += int(stats[0])
    return got, prev, st, seg


@pytest.mark.parametrize("traversal", ["brute", "bvh", "bvh_ordered"])
@pytest.mark.parametrize("mode", list(range(9)) + [10, -3])
def test_every_integrator_mode(native, oracle, traversal, mode):
    """eval_integrator modes 0..8 and the default branch (integrator_Hart) full screen on the mirror/glass/emitter
    scene (integrators.glsl:24-543, 681-693)."""
    from rvpt_amd import Camera
    sc = scene_by_name("showcase")
    c = Camera(128 / 80)
    c.translation = np.array([0.3, 1.1, -2.2])
    c.rotation = np.array([-8.0, 6.0, 0.0])
    kw = dict(aa=2, max_bounces=5, top_left_render_mode=mode, top_right_render_mode=mode, bottom_left_render_mode=mode,
              bottom_right_render_mode=mode)
    got, ref, st, seg = _frames_with_settings(native, oracle, sc, c.get_data(), 128, 80, traversal, kw)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"mode {mode} {traversal}: {int((got != ref).any(axis=2).sum())} pixels differ"
    assert st[0] == seg


@pytest.mark.parametrize("traversal", ["brute", "bvh"])
def test_split_screen_and_cameras(native, oracle, traversal):
    """compute_pass.comp:134-144 split screen with four different integrators; ortho and spherical cameras
    (camera.glsl:55-99) under Kajiya and under the depth view."""
    from rvpt_amd import Camera
    sc = scene_by_name("showcase")
    c = Camera(144 / 96)
    c.translation = np.array([0.2, 1.3, -2.4])
    c.rotation = np.array([5.0, 8.0, 0.0])
    cam = c.get_data()
    kw = dict(aa=2, top_left_render_mode=7, top_right_render_mode=5, bottom_left_render_mode=8, bottom_right_render_mode=9,
              split_ratio=(0.4, 0.6), max_bounces=4)
    got, ref, st, seg = _frames_with_settings(native, oracle, sc, cam, 144, 96, traversal, kw)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert st[0] == seg
    for cam_mode in (1, 2, 7):
        for mode in (9, 2):
            kw = dict(aa=1, camera_mode=cam_mode, top_left_render_mode=mode, top_right_render_mode=mode, bottom_left_render_mode=mode,
                      bottom_right_render_mode=mode)
            got, ref, _, _ = _frames_with_settings(native, oracle, sc, cam, 96, 64, traversal, kw)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (cam_mode, mode)


def test_ambient_occlusion_with_zero_rays_is_nan_like_upstream(native, oracle):
    """integrator_ao with nrays = max_bounces = 0 evaluates 1 - 0/0 on hit pixels (integrators.glsl:191-207)."""
    sc = scene_by_name("default")
    kw = dict(max_bounces=0, top_left_render_mode=5, top_right_render_mode=5, bottom_left_render_mode=5, bottom_right_render_mode=5)
    got, ref, _, _ = _frames_with_settings(native, oracle, sc, identity_camera(1.0), 32, 32, "brute", kw, frames=(0,))
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(got).any()
    assert np.array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])


def test_converged_mean_agrees_with_an_independent_estimator(native, oracle):
    """two estimators with disjoint random streams converge to the same image.  GPU: 256 accumulated
    frames x 4 spp (seeds hash(pixel)+0..255); oracle: one frame with 1024 spp (a single long stream per pixel)."""
    sc = scene_by_name("default")
    tris, mats, nodes = sc
    W, H = 32, 32
    cam = identity_camera(1.0)
    from rvpt_amd import RenderSettings
    ctx = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        ctx.upload_scene(nodes, tris, mats)
        for f in range(256):
            ctx.set_frame(RenderSettings(aa=4, current_frame=f).pack(), cam)
            ctx.dispatch()
        gpu = ctx.read()[..., :3].astype(np.float64)
    finally:
        ctx.close()
    ref, _ = oracle.render(oracle.settings_bytes(aa=1024, current_frame=0), cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BVH)
    ref = ref[..., :3].astype(np.float64)
    assert abs(gpu.mean() - ref.mean()) / ref.mean() < 0.01
    assert np.sqrt(((gpu - ref) ** 2).mean()) < 0.05  # per-pixel Monte-Carlo error at 1024 samples


@pytest.mark.parametrize("cells", [21, 22, 23])
def test_brute_force_around_the_lds_capacity_boundary(native, oracle, cells):
    """882 / 968 / 1058 triangles: the last sizes that stay LDS-resident and the first that stream (64 KiB of LDS
    per work-group: 64 B record + 4 B material index per triangle + materials + the split-mode table)."""
    from rvpt_amd import Camera, scene
    tris, mats = scene.heightfield_scene(cells=cells)
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    c = Camera(48 / 32)
    c.translation = np.array([0.0, 2.5, -5.0])
    c.rotation = np.array([0.0, 25.0, 0.0])
    got, _ = gpu_frames(native, sc, c.get_data(), 48, 32, "brute", [0, 1], aa=2)
    ref, _ = oracle_frames(oracle, sc, c.get_data(), 48, 32, "brute", [0, 1], aa=2)
    assert np.array_equal(got[1], ref[1])


def test_scene_change_and_interleaved_contexts(native, oracle):
    """upload_scene while frames are in flight (must drain them first), and two contexts used alternately."""
    from rvpt_amd import RenderSettings
    a, b = scene_by_name("default"), scene_by_name("showcase")
    W, H = 64, 48
    cam = identity_camera(W / H)
    c1 = native.Context(W, H, 0, 0, 1, 0)
    c2 = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        c1.upload_scene(None, a[0], a[1])
        c2.upload_scene(b[2], b[0], b[1])
        for f in range(4):
            for c in (c1, c2):
                c.set_frame(RenderSettings(current_frame=f).pack(), cam)
                c.dispatch()
        c1.upload_scene(None, b[0], b[1])  # scene swap with 4 frames queued: they must finish on the old scene
        assert c1.query() is False
        img_a = c1.read()
        for f in range(3):
            c1.set_frame(RenderSettings(current_frame=f).pack(), cam)
            c1.dispatch()
        img_b1, img_b2 = c1.read(), c2.read()
    finally:
        c1.close()
        c2.close()
    ref_a, _ = oracle_frames(oracle, a, cam, W, H, "brute", [0, 1, 2, 3])
    ref_b1, _ = oracle_frames(oracle, b, cam, W, H, "brute", [0, 1, 2])
    ref_b2, _ = oracle_frames(oracle, b, cam, W, H, "bvh", [0, 1, 2, 3])
    assert np.array_equal(img_a, ref_a[-1]) and np.array_equal(img_b1, ref_b1[-1]) and np.array_equal(img_b2, ref_b2[-1])


def test_randomised_sweep(native, oracle):
    """60 random combinations of size, spp, bounces, per-quadrant integrators, split, camera mode/pose, traversal,
    kernel flavour, tile partition and scene (tools/fuzz_parity.py; 1300 such cases were run when it was written)."""
    import importlib.util
    from _util import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_parity", ROOT / "tools" / "fuzz_parity.py")
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(60, 2024) == 0


@pytest.mark.parametrize("scene_name,W,H", [("default", 160, 96), ("showcase", 128, 80), ("cornell", 96, 64)])
def test_ordered_bvh_traversal(native, oracle, scene_name, W, H):
    """RVPT_HIP_TRAVERSAL_BVH_ORDERED (nearer child first — the reference's TODO at intersection.glsl:405): bit-exact
    against the oracle's ordered variant, and the same image as the reference order except for tie / slab-rounding pixels."""
    from rvpt_amd import Camera
    sc = scene_by_name(scene_name)
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9]) if scene_name == "cornell" else np.array([0.2, 1.0, -2.3])
    cam = c.get_data()
    got, st = gpu_frames(native, sc, cam, W, H, "bvh_ordered", [0, 1, 2], aa=2, flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, W, H, "bvh_ordered", [0, 1, 2], aa=2)
    assert np.array_equal(got[2], ref[2]) and st[0] == seg
    plain, _ = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1, 2], aa=2)
    assert int((plain[2] != got[2]).any(axis=2).sum()) <= 1e-3 * W * H


def test_ordered_bvh_million_triangles(native, oracle):
    from rvpt_amd import Camera, scene
    tris, mats = scene.heightfield_scene()
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    c = Camera(64 / 40)
    c.translation = np.array([0.0, 2.5, -5.0])
    c.rotation = np.array([0.0, 25.0, 0.0])
    got, _ = gpu_frames(native, sc, c.get_data(), 64, 40, "bvh_ordered", [0, 1], aa=2)
    ref, _ = oracle_frames(oracle, sc, c.get_data(), 64, 40, "bvh_ordered", [0, 1], aa=2)
    assert np.array_equal(got[1], ref[1])


def _chain_bvh(tris):
    """A maximally unbalanced tree over `tris` (already in leaf order): inner node k = {leaf(tri k), rest}."""
    n = tris.shape[0]
    v = tris.reshape(n, 4, 4)[:, :3, :3]
    lo, hi = v.min(axis=1), v.max(axis=1)
    nodes = np.zeros(2 * n - 1, dtype=np.dtype([("first", "<u4"), ("count", "<u4"), ("bounds", "<f4", (6,))]))

    def box(a, b):
        l, h = lo[a:b].min(axis=0), hi[a:b].max(axis=0)
        return [l[0], h[0], l[1], h[1], l[2], h[2]]

    at = 0  # node holding triangles [k, n)
    for k in range(n - 1):
        nodes[at] = (2 * k + 1, 0, box(k, n))
        nodes[2 * k + 1] = (k, 1, box(k, k + 1))
        at = 2 * k + 2
    nodes[at] = (n - 1, 1, box(n - 1, n))
    return nodes


@pytest.mark.parametrize("n_quads", [24, 32])
@pytest.mark.parametrize("traversal", ["bvh", "bvh_ordered"])
def test_deep_chain_tree_stack(native, oracle, traversal, n_quads):
    """Tree height 48 and 64 (the deepest tree the reference's 64-entry stack walks: sentinel + 63 pushes): the traversal
    stack is sized from the tree (ordered: 8 bytes per level and lane -> more than the default 64 KiB of LDS per work-group)."""
    from rvpt_amd import Camera, scene
    rng = np.random.RandomState(11)
    quads = []
    for k in range(n_quads):
        z = 1.0 + 0.25 * k
        s = 0.3 + 0.05 * k
        c = rng.uniform(-0.5, 0.5, 2)
        p = [(c[0] - s, c[1] - s, z), (c[0] + s, c[1] - s, z), (c[0] + s, c[1] + s, z), (c[0] - s, c[1] + s, z)]
        quads += [(p[0], p[1], p[2], k % 3), (p[0], p[2], p[3], k % 3)]
    tris = scene.make_triangles([q[:3] for q in quads], 0)
    tris[:, 12] = np.asarray([q[3] for q in quads], dtype=np.float32)
    mats = np.stack([scene.make_material((0.9, 0.9, 0.9, 0), (0.1, 0.4, 0.6, 0), 0), scene.make_material((0.8, 0.3, 0.3, 0), (0, 0, 0, 0), 1),
                     scene.make_material((1, 1, 1, 1.5), (0, 0, 0, 0), 2)])
    nodes = _chain_bvh(tris)
    sc = (tris, mats, nodes)
    cam = Camera(96 / 64).get_data()
    got, st = gpu_frames(native, sc, cam, 96, 64, traversal, [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 96, 64, traversal, [0, 1], aa=2)
    assert np.array_equal(got[1], ref[1]) and st[0] == seg
    if n_quads == 32:  # one level more than the reference's stack can walk: rejected at upload, not rendered wrongly
        t65 = np.concatenate([tris, tris[:1]])
        ctx = native.Context(32, 32, 0, 0, 1, native.TRAVERSAL_BVH)
        try:
            with pytest.raises(native.NativeError, match="BVH height 65"):
                ctx.upload_scene(_chain_bvh(t65), t65, mats)
        finally:
            ctx.close()


@pytest.mark.parametrize("fetch_heads", [False, True])
def test_popped_node_heads_packed_on_the_stack_or_fetched(native, oracle, monkeypatch, fetch_heads):
    """A stacked child carries its (first, count) pair packed into its slot; trees whose leaf sizes do not fit beside their indices
    (and RVPT_HIP_BVH_NO_PACKED_HEADS) keep the node index there and fetch the pair at the pop.  Both walk the reference's order."""
    from rvpt_amd import Camera, scene
    if fetch_heads:
        monkeypatch.setenv("RVPT_HIP_LAB", "1")  # a knob of the laboratory build (include/rvpt_hip_lab.h)
        monkeypatch.setenv("RVPT_HIP_BVH_NO_PACKED_HEADS", "1")
    tris, mats = scene.cornell_scene()
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    c = Camera(64 / 48)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()
    for traversal in ("bvh", "bvh_ordered"):
        got, st = gpu_frames(native, sc, cam, 64, 48, traversal, [0, 1], aa=1, flags=native.COUNT_SEGMENTS)
        ref, seg = oracle_frames(oracle, sc, cam, 64, 48, traversal, [0, 1], aa=1)
        assert np.array_equal(got[1], ref[1]) and st[0] == seg


CAMPACK_KNOBS = [  # (RVPT_HIP_BVH_CAM_MIN, RVPT_HIP_BVH_DETACH)
    (None, None),  # the built-in policy
    (1, 0),        # every group of fresh lanes walks as a packet, down to the leaves (nobody ever leaves)
    (1, 63),       # ... and leaves it at the first node that does not hold the whole wave
    (2, 1), (16, 8), (64, 31),
]


@pytest.mark.parametrize("scene_name,W,H,cam_t", [("default", 160, 96, (0.2, 0.9, -2.4)), ("showcase", 96, 64, (0.0, 1.2, -3.0))])
@pytest.mark.parametrize("cam_min,detach", CAMPACK_KNOBS)
def test_camera_packets_equal_the_per_lane_walk(native, oracle, monkeypatch, scene_name, W, H, cam_t, cam_min, detach):
    """Camera packets (trace_bvh4_resident: lanes that start camera rays together walk the wide tree as one wave-uniform packet in the reference's fixed
    child order, intersection.glsl:361-413) against the binary per-lane walk (RVPT_HIP_BVH_PER_LANE) and the oracle: images AND segment counts bit for bit,
    for every packet size / detach threshold — a lane's sequence of passed boxes and tested triangles is the one it walks alone, whenever it leaves the
    packet.  aa = 2 (a pixel's second sample joins later packets), two frames.  (The binary-tree form of the packet walk, LDS- and HBM-resident, passed the same
    matrix before it was retired: profiles/r04_exp_campack_binary.patch.)"""
    from rvpt_amd import Camera
    if cam_min is not None:
        monkeypatch.setenv("RVPT_HIP_LAB", "1")  # a knob of the laboratory build (include/rvpt_hip_lab.h)
        monkeypatch.setenv("RVPT_HIP_BVH_CAM_MIN", str(cam_min))
        monkeypatch.setenv("RVPT_HIP_BVH_DETACH", str(detach))
    sc = scene_by_name(scene_name)
    c = Camera(W / H)
    c.translation = np.array(cam_t)
    cam = c.get_data()
    got, st = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    lane, st_lane = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS | native.BVH_PER_LANE)
    ref, seg = oracle_frames(oracle, sc, cam, W, H, "bvh", [0, 1], aa=2)
    for f in range(2):
        assert np.array_equal(got[f].view(np.uint32), lane[f].view(np.uint32)), f"frame {f}: camera packets != per-lane walk"
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f"frame {f}: camera packets != oracle"
    assert st == st_lane and st[0] == seg


def test_camera_packet_kernel_is_what_bvh_contexts_run(native):
    """The default policy: BVH contexts in the lean configuration (Kajiya, pinhole, reference order) walk the 4-wide tree — with the scene and camera
    packets in LDS when it fits (variant 11), through L2 when it does not (10) — the other render modes too (GENERIC instances of the same kernels);
    RVPT_HIP_BVH_PER_LANE and the nearer-child-first order keep the binary per-lane kernels (3 / 2)."""
    from rvpt_amd import RenderSettings
    for name, want, plain in (("default", 11, 3), ("cornell", 10, 2)):
        tris, mats, nodes = scene_by_name(name)
        for flags, mode, expect in ((native.TRAVERSAL_BVH, 9, want), (native.TRAVERSAL_BVH | native.BVH_PER_LANE, 9, plain),
                                    (native.TRAVERSAL_BVH_ORDERED, 9, plain), (native.TRAVERSAL_BVH, 4, want)):
            ctx = native.Context(64, 32, 0, 0, 1, flags)
            try:
                ctx.upload_scene(nodes, tris, mats)
                rs = RenderSettings(max_bounces=4, aa=1, current_frame=0)
                rs.top_left_render_mode = rs.top_right_render_mode = rs.bottom_left_render_mode = rs.bottom_right_render_mode = mode
                ctx.set_frame(rs.pack(), identity_camera(2.0))
                ctx.dispatch()
                ctx.wait()
                assert ctx.launch_info()[2] == expect, (name, flags, mode, ctx.launch_info())
            finally:
                ctx.close()


def _loosen_boxes(nodes, seed, frac=0.3):
    """Inner boxes of a valid tree shrunk or grown at random, so that many of them no longer contain their children (a caller's tree
    may be like that; the reference then culls with the boxes it is given, intersection.glsl:377-380, and so must every kernel)."""
    from rvpt_amd import native as nat
    rng = np.random.RandomState(seed)
    out = np.ascontiguousarray(nodes).view(nat.NODE_DTYPE).reshape(-1).copy()  # build_bvh returns uint32[n, 8]
    inner = np.flatnonzero(out["count"] == 0)
    pick = inner[rng.rand(inner.size) < frac]
    b = out["bounds"][pick].astype(np.float64)
    centre = (b[:, 0::2] + b[:, 1::2]) / 2
    half = (b[:, 1::2] - b[:, 0::2]) / 2 * rng.uniform(0.55, 1.3, (pick.size, 3))
    nb = np.empty_like(b)
    nb[:, 0::2], nb[:, 1::2] = centre - half, centre + half
    out["bounds"][pick] = nb.astype(np.float32)
    return out


@pytest.mark.parametrize("loose", [False, True])
def test_wide_tree_walk_equals_the_binary_walk(native, oracle, monkeypatch, loose):
    """rvpt_bvh4.hip walks the 4-wide regrouping of the caller's tree (rvpt_abi.hip: build_wide_nodes) and must find what the reference's
    binary walk finds, bit for bit, segment counts included: nodes are regrouped only across boxes that contain their children, so the
    reference's rule — a node is visited iff its own box passes when the depth-first order reaches it — is kept.  `loose`: a third of the
    inner boxes shrunk / grown at random (many no longer contain their children; rays are culled where the reference would cull them)."""
    from rvpt_amd import Camera, scene
    tris, mats = scene.cornell_scene()
    nodes, idx = native.build_bvh(tris)
    if loose:
        nodes = _loosen_boxes(nodes, 5)
    sc = (tris[idx], mats, nodes)
    W, H = 128, 80
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()
    ref, seg = oracle_frames(oracle, sc, cam, W, H, "bvh", [0, 1], aa=2)
    got, st = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    lane, st_lane = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS | native.BVH_PER_LANE)
    for f in range(2):
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f"frame {f}: wide walk != oracle"
        assert np.array_equal(lane[f].view(np.uint32), ref[f].view(np.uint32)), f"frame {f}: binary walk != oracle"
    assert st == st_lane and st[0] == seg
    if loose:  # the loosened boxes do cull: the image is NOT the valid tree's
        valid, _ = oracle_frames(oracle, (tris[idx], mats, native.build_bvh(tris)[0]), cam, W, H, "bvh", [0])
        assert not np.array_equal(valid[0], ref[0])
    # the kernel that ran: the wide one by default, the binary one on request and for the other configurations
    ctx = native.Context(64, 32, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        from rvpt_amd import RenderSettings
        ctx.upload_scene(nodes, tris[idx], mats)
        ctx.set_frame(RenderSettings(max_bounces=2, aa=1, current_frame=0).pack(), cam)
        ctx.dispatch()
        ctx.wait()
        assert ctx.launch_info()[2] == 10
    finally:
        ctx.close()


@pytest.mark.parametrize("which", ["cornell", "terrain", "loose"])
def test_quantised_node_walk_equals_the_reference_walk(native, oracle, monkeypatch, which):
    """RVPT_HIP_BVH_QUANT=1 (trace_bvh4q, opt-in: bit-exact and measured slower, profiles/EXPERIMENTS.md 5.16): the 4-wide walk over 64-byte nodes whose child
    boxes are 8-bit supersets, tested with a conservative slab test, a leaf's own box tested exactly at its visit.  Inner boxes only cull (containment),
    so images AND segment counts are the reference's; a tree whose boxes do not contain their children has no quantised form and keeps the exact nodes."""
    from rvpt_amd import Camera, RenderSettings, scene
    monkeypatch.setenv("RVPT_HIP_LAB", "1")  # a knob of the laboratory build (include/rvpt_hip_lab.h)
    monkeypatch.setenv("RVPT_HIP_BVH_QUANT", "1")
    if which == "terrain":
        tris, mats = scene.heightfield_scene(64)
        pos, rot = np.array([0.0, 2.5, -5.0]), np.array([0.0, 25.0, 0.0])
    else:
        tris, mats = scene.cornell_scene()
        pos, rot = np.array([0.0, 2.0, -1.9]), np.zeros(3)
    nodes, idx = native.build_bvh(tris)
    if which == "loose":
        nodes = _loosen_boxes(nodes, 5)
    sc = (tris[idx], mats, nodes)
    W, H = 128, 80
    c = Camera(W / H)
    c.translation, c.rotation = pos, rot
    cam = c.get_data()
    ref, seg = oracle_frames(oracle, sc, cam, W, H, "bvh", [0, 1], aa=2)
    got, st = gpu_frames(native, sc, cam, W, H, "bvh", [0, 1], aa=2, flags=native.COUNT_SEGMENTS)
    for f in range(2):
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f"frame {f}: quantised walk != oracle"
    assert st[0] == seg
    ctx = native.Context(64, 32, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        ctx.upload_scene(nodes, tris[idx], mats)
        ctx.set_frame(RenderSettings(max_bounces=2, aa=1, current_frame=0).pack(), cam)
        ctx.dispatch()
        ctx.wait()
        assert ctx.launch_info()[2] == (10 if which == "loose" else 13)
    finally:
        ctx.close()


def test_a_lone_launch_of_many_frames_takes_the_whole_cu(native):
    """choose_launch: a launch of >= 16 frames of the HBM-resident BVH kernels that goes out while nothing of the context is in flight (a rank's K-step share
    sent as one launch) takes what the registers allow instead of the three work-groups per CU that leave room for launches in flight; launches that follow
    while it runs, and streams of smaller launches, keep the overlapping shape.  Same image either way (the grid only changes who renders which pixel)."""
    from rvpt_amd import Camera, RenderSettings
    tris, mats, nodes = scene_by_name("cornell")
    W, H = 512, 288
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()

    def run(plan):
        ctx = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BVH)
        grids = []
        try:
            ctx.upload_scene(nodes, tris, mats)
            f = 0
            for n, wait_first in plan:
                if wait_first:
                    ctx.wait()
                ctx.set_frame(RenderSettings(max_bounces=4, aa=1, current_frame=f).pack(), cam)
                ctx.dispatch_frames(n)
                grids.append(ctx.launch_info()[0])
                f += n
            return ctx.read(), grids
        finally:
            ctx.close()

    lone, g_lone = run([(16, True), (16, True)])          # each launch finds the context idle
    small, g_small = run([(8, True), (8, True), (8, True), (8, True)])  # lone but < 16 frames: the overlapping shape
    assert np.array_equal(lone.view(np.uint32), small.view(np.uint32))
    assert g_lone[0] == g_lone[1] and g_small[0] == g_small[1] and g_lone[0] > g_small[0], (g_lone, g_small)


@pytest.mark.parametrize("scene_name,flags", [("cornell", 0), ("cornell", "per_lane"), ("default", 0)])
def test_a_traversal_stack_that_is_too_small_is_reported_not_silent(native, monkeypatch, scene_name, flags):
    """The BVH kernels clamp a push at the top of the stack the host sized — like the reference's uint stack[64] they would otherwise run past it
    (intersection.glsl:367).  In the DEBUG build of the library, with RVPT_HIP_DEBUG=1, a clamped push is an ERROR at rvpt_hip_wait, not a silently wrong frame: forced here by lying to the kernels
    about the levels the tree needs (wide walk, binary per-lane walk, LDS-resident wide walk); with the true bound the same run reports nothing."""
    from rvpt_amd import Camera, RenderSettings
    W, H = 160, 96
    tris, mats, nodes = scene_by_name(scene_name)
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9]) if scene_name == "cornell" else np.array([0.0, 0.9, -2.5])
    fl = native.TRAVERSAL_BVH | (native.BVH_PER_LANE if flags == "per_lane" else 0)
    # the DEBUG build of the library carries the check (the release kernels only clamp: the never-taken branch measured -3.4 % on C3); a subprocess, because a
    # process loads one library
    import subprocess, sys, textwrap
    from rvpt_amd import build
    build.build_native_debug()
    env = dict(os.environ, RVPT_HIP_LAB="1", RVPT_HIP_DEBUG="1")
    code = textwrap.dedent(f"""
        import sys, numpy as np
        sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / "tests")!r})
        import os
        from rvpt_amd import native, Camera, RenderSettings
        from _util import scene_by_name
        W, H = {W}, {H}
        tris, mats, nodes = scene_by_name({scene_name!r})
        c = Camera(W / H)
        c.translation = np.array({[float(x) for x in c.translation]!r})
        outcome = []
        for forced in (None, "1"):
            if forced:
                os.environ["RVPT_HIP_BVH_FORCE_STACK_LEVELS"] = forced
            ctx = native.Context(W, H, 0, 0, 1, {fl})
            ctx.upload_scene(nodes, tris, mats)
            ctx.set_frame(RenderSettings(aa=2, current_frame=0).pack(), c.get_data())
            ctx.dispatch()
            try:
                ctx.wait()
                outcome.append("ok")
            except native.NativeError as e:
                outcome.append("stack overflow" if "stack overflow" in str(e) else str(e))
                ctx.wait()  # the word is cleared by the report: reported once
            ctx.close()
        print("OUTCOME", outcome)
    """)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "OUTCOME ['ok', 'stack overflow']" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_the_tree_top_knob_is_clamped_to_what_a_work_group_can_hold(native, oracle, monkeypatch):
    """RVPT_HIP_BVH_TOP_NODES counts nodes of whatever tree the kernel walks; 2048 wide nodes would be 256 KiB of LDS: the launcher clamps the
    request to the 64 KiB a work-group can have, and the frame is the oracle's as with any other value of the knob (0: no LDS copy at all)."""
    from rvpt_amd import Camera
    sc = scene_by_name("cornell")
    c = Camera(96 / 64)
    c.translation = np.array([0.0, 2.0, -1.9])
    ref, _ = oracle_frames(oracle, sc, c.get_data(), 96, 64, "bvh", [0, 1], aa=2)
    for knob in ("2048", "400", "0"):
        monkeypatch.setenv("RVPT_HIP_LAB", "1")  # a knob of the laboratory build (include/rvpt_hip_lab.h)
        monkeypatch.setenv("RVPT_HIP_BVH_TOP_NODES", knob)
        got, _ = gpu_frames(native, sc, c.get_data(), 96, 64, "bvh", [0, 1], aa=2)
        assert_parity(got[1], ref[1], f"top nodes {knob}", max_mismatch_frac=0)


def test_unknown_create_flags_are_rejected(native):
    """ABI 5: the wavefront pipelines are retired; their flag bits (0x40, 0x80, 0x100) and any other unknown bit fail at create."""
    for bad in (0x40, 0x80, 0x100, 0x800, 1 << 31):
        with pytest.raises(native.NativeError, match="unknown bits"):
            native.Context(32, 32, 0, 0, 1, native.TRAVERSAL_BVH | bad)


@pytest.mark.parametrize("caller_layout", [False, True])
def test_caller_node_layout_and_a_single_leaf_tree(native, oracle, monkeypatch, caller_layout):
    """RVPT_HIP_BVH_CALLER_LAYOUT keeps the uploaded node order (sibling pairs wherever the caller put them, not on 64-byte lines);
    a tree that is one leaf has no pair at all.  Same images either way."""
    from rvpt_amd import Camera, scene
    if caller_layout:
        monkeypatch.setenv("RVPT_HIP_LAB", "1")  # a knob of the laboratory build (include/rvpt_hip_lab.h)
        monkeypatch.setenv("RVPT_HIP_BVH_CALLER_LAYOUT", "1")
    c = Camera(64 / 48)
    c.translation = np.array([0.0, 2.0, -1.9])
    tris, mats = scene.cornell_scene()  # HBM-resident kernel
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    got, _ = gpu_frames(native, sc, c.get_data(), 64, 48, "bvh", [0, 1])
    ref, _ = oracle_frames(oracle, sc, c.get_data(), 64, 48, "bvh", [0, 1])
    assert np.array_equal(got[1], ref[1])
    tris, mats = scene.default_scene()  # LDS-resident kernel
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    cam = Camera(64 / 48).get_data()
    got, _ = gpu_frames(native, sc, cam, 64, 48, "bvh_ordered", [0, 1])
    ref, _ = oracle_frames(oracle, sc, cam, 64, 48, "bvh_ordered", [0, 1])
    assert np.array_equal(got[1], ref[1])
    three = tris[idx][:3].copy()  # one leaf = the whole tree
    v = three.reshape(3, 4, 4)[:, :3, :3]
    lo, hi = v.min(axis=(0, 1)), v.max(axis=(0, 1))
    root = np.zeros(1, dtype=np.dtype([("first", "<u4"), ("count", "<u4"), ("bounds", "<f4", (6,))]))
    root[0] = (0, 3, [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]])
    sc = (three, mats, root)
    for traversal in ("bvh", "bvh_ordered"):
        got, _ = gpu_frames(native, sc, cam, 64, 48, traversal, [0])
        ref, _ = oracle_frames(oracle, sc, cam, 64, 48, traversal, [0])
        assert np.array_equal(got[0], ref[0])


def test_leaf_too_large_to_pack_beside_the_indices(native, oracle):
    """66 000 triangles, one leaf of 40 000: 17 bits of index leave 15 for a leaf size, 40 000 does not fit -> the host must choose the
    fetching pop (packing it anyway would corrupt the traversal and this image)."""
    from rvpt_amd import Camera, scene
    n_side = 182  # 182 x 182 cells x 2 triangles = 66 248
    xs = np.linspace(-1.5, 1.5, n_side + 1, dtype=np.float32)
    corners = []
    for j in range(n_side):
        for i in range(n_side):
            z = np.float32(3.0 + 0.2 * np.sin(0.7 * i) * np.cos(0.5 * j))
            p00, p10, p11, p01 = (xs[i], xs[j], z), (xs[i + 1], xs[j], z), (xs[i + 1], xs[j + 1], z), (xs[i], xs[j + 1], z)
            corners += [(p00, p10, p11), (p00, p11, p01)]
    tris = scene.make_triangles(corners, 0)
    tris[::3, 12] = 1.0
    mats = np.stack([scene.make_material((0.8, 0.8, 0.8, 0), (0.2, 0.3, 0.1, 0), 0), scene.make_material((0.9, 0.9, 0.9, 0), (0, 0, 0, 0), 1)])
    n = tris.shape[0]
    v = tris.reshape(n, 4, 4)[:, :3, :3]
    nodes = np.zeros(3, dtype=np.dtype([("first", "<u4"), ("count", "<u4"), ("bounds", "<f4", (6,))]))

    def box(a, b):
        l, h = v[a:b].min(axis=(0, 1)), v[a:b].max(axis=(0, 1))
        return [l[0], h[0], l[1], h[1], l[2], h[2]]

    nodes[0] = (1, 0, box(0, n))
    nodes[1] = (0, 40000, box(0, 40000))
    nodes[2] = (40000, n - 40000, box(40000, n))
    sc = (tris, mats, nodes)
    cam = Camera(32 / 16).get_data()
    got, st = gpu_frames(native, sc, cam, 32, 16, "bvh", [0], aa=1, max_bounces=3, flags=native.COUNT_SEGMENTS)
    ref, seg = oracle_frames(oracle, sc, cam, 32, 16, "bvh", [0], aa=1, max_bounces=3)
    assert np.array_equal(got[0], ref[0]) and st[0] == seg


def test_empty_scene_in_a_bvh_context_is_the_sky(native, oracle):
    """RVPT::initialize() with no triangles and the (default) BVH traversal: no tree exists; every ray misses."""
    from rvpt_amd import RenderSettings
    cam = identity_camera(2.0)
    mats = np.zeros((1, 12), np.float32)
    tris = np.zeros((0, 16), np.float32)
    ctx = native.Context(64, 32, 0, 0, 1, native.TRAVERSAL_BVH)
    try:
        ctx.upload_scene(None, tris, mats)
        ctx.set_frame(RenderSettings(aa=2, current_frame=0).pack(), cam)
        ctx.dispatch()
        got = ctx.read()
    finally:
        ctx.close()
    ref, _ = oracle.render(oracle.settings_bytes(aa=2, current_frame=0), cam, None, tris, mats, 64, 32, oracle.TRAVERSAL_BRUTE)
    assert np.array_equal(got, ref)


def _batched(native, sc, cam, W, H, traversal, plan, flags=0, world=1, rank=0, aa=2, modes=(9, 9, 9, 9), camera_mode=0):
    """plan = [(first_frame, n_frames), ...]: one rvpt_hip_dispatch_frames() call per entry."""
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    fl = flags | native.COUNT_SEGMENTS | {"bvh": native.TRAVERSAL_BVH, "brute": native.TRAVERSAL_BRUTE, "bvh_ordered": native.TRAVERSAL_BVH_ORDERED}[traversal]
    ctx = native.Context(W, H, 0, rank, world, fl)
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        for first, n in plan:
            rs = RenderSettings(max_bounces=6, aa=aa, current_frame=first, camera_mode=camera_mode, top_left_render_mode=modes[0],
                                top_right_render_mode=modes[1], bottom_left_render_mode=modes[2], bottom_right_render_mode=modes[3])
            ctx.set_frame(rs.pack(), cam)
            if n == 1:
                ctx.dispatch()
            else:
                ctx.dispatch_frames(n)
        return ctx.read(), ctx.stats()
    finally:
        ctx.close()


@pytest.mark.parametrize("traversal", ["brute", "bvh", "bvh_ordered"])
def test_dispatch_frames_equals_frame_by_frame(native, oracle, traversal):
    """rvpt_hip_dispatch_frames(n): one launch over n consecutive frames == n dispatches, bit for bit (and == the oracle)."""
    from rvpt_amd import Camera
    W, H = 112, 72  # partial edge tiles
    sc = scene_by_name("showcase")
    c = Camera(W / H)
    c.translation = np.array([0.2, 1.0, -2.3])
    cam = c.get_data()
    single, st1 = _batched(native, sc, cam, W, H, traversal, [(f, 1) for f in range(11)])
    batched, st2 = _batched(native, sc, cam, W, H, traversal, [(0, 4), (4, 1), (5, 6)])
    assert np.array_equal(single, batched) and tuple(st1) == tuple(st2)
    ref, _ = oracle_frames(oracle, sc, cam, W, H, traversal, list(range(11)), aa=2, max_bounces=6)
    assert np.array_equal(batched, ref[-1])
    # reference-format accumulation quantises after every frame, also inside a batch; tile partitions; generic kernels
    q1, _ = _batched(native, sc, cam, W, H, traversal, [(f, 1) for f in range(6)], flags=native.ACCUM_UNORM8)
    q2, _ = _batched(native, sc, cam, W, H, traversal, [(0, 6)], flags=native.ACCUM_UNORM8)
    assert np.array_equal(q1, q2)
    t1, _ = _batched(native, sc, cam, W, H, traversal, [(f, 1) for f in range(5)], world=3, rank=1)
    t2, _ = _batched(native, sc, cam, W, H, traversal, [(0, 3), (3, 2)], world=3, rank=1)
    assert np.array_equal(t1, t2) and t2.any()
    g1, _ = _batched(native, sc, cam, W, H, traversal, [(f, 1) for f in range(4)], modes=(5, 7, 8, 9), camera_mode=2)
    g2, _ = _batched(native, sc, cam, W, H, traversal, [(0, 4)], modes=(5, 7, 8, 9), camera_mode=2)
    assert np.array_equal(g1, g2, equal_nan=True)


@pytest.mark.parametrize("scene_name", ["default", "showcase"])
def test_packet_kernel_equals_the_mixed_packet_kernel(native, scene_name):
    """The default brute-force kernel for LDS-resident scenes in the lean configuration (rvpt_packets.hip: camera rounds with the
    packet-uniform early-out, bounce rounds fed from a 64-entry LDS queue, split mode in the tail) against round 2's
    trace_brute_resident (RVPT_HIP_BRUTE_MIXED_PACKETS): same image, same segment and sample counts — partial edge tiles, 1-3 spp,
    1 and 8 bounces, frames one by one and in batches, a 3-way tile partition, rgba8 accumulation."""
    from rvpt_amd import Camera
    W, H = 208, 120
    sc = scene_by_name(scene_name)
    c = Camera(W / H)
    c.translation = np.array([0.15, 0.95, -2.35])
    c.rotation = np.array([3.0, -8.0, 0.0])
    cam = c.get_data()
    tris, mats, nodes = sc

    def run(flags, plan, aa, bounces, world=1, rank=0):
        from rvpt_amd import RenderSettings
        ctx = native.Context(W, H, 0, rank, world, flags | native.COUNT_SEGMENTS)
        try:
            ctx.upload_scene(None, tris, mats)
            for first, n in plan:
                ctx.set_frame(RenderSettings(max_bounces=bounces, aa=aa, current_frame=first).pack(), cam)
                ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
            img = ctx.read(native.FORMAT_RGBA8_UNORM) if (flags & native.ACCUM_UNORM8) else ctx.read()
            return img, ctx.stats(), ctx.launch_info()[2]
        finally:
            ctx.close()

    for plan, aa, bounces, world, rank, extra in (([(f, 1) for f in range(5)], 1, 8, 1, 0, 0),
                                                  ([(0, 3), (3, 1), (4, 6)], 3, 8, 1, 0, 0),
                                                  ([(0, 4)], 2, 1, 1, 0, 0),
                                                  ([(0, 2), (2, 3)], 2, 8, 3, 2, 0),
                                                  ([(f, 1) for f in range(4)], 1, 8, 1, 0, native.ACCUM_UNORM8)):
        new, st_new, v_new = run(extra, plan, aa, bounces, world, rank)
        old, st_old, v_old = run(extra | native.BRUTE_MIXED_PACKETS, plan, aa, bounces, world, rank)
        assert (v_new, v_old) == (6, 0)
        assert np.array_equal(new.view(np.uint8), old.view(np.uint8)), (plan, aa, bounces, world)
        assert tuple(st_new) == tuple(st_old)
        assert new.any()
    # a generic render mode or a non-pinhole camera stays on the generic instance of the round-2 kernel
    from rvpt_amd import RenderSettings
    ctx = native.Context(W, H, 0, 0, 1, 0)
    try:
        ctx.upload_scene(None, tris, mats)
        ctx.set_frame(RenderSettings(aa=1, current_frame=0, bottom_right_render_mode=3).pack(), cam)
        ctx.dispatch()
        ctx.wait()
        assert ctx.launch_info()[2] == 0
    finally:
        ctx.close()


RECT_CAMERAS = [((0, 0, 0), (0, 0, 0), 90.0), ((0, 0.9, -2.5), (0, 0, 0), 90.0), ((1.4, 1.6, -1.2), (-40.0, 25.0, 10.0), 70.0), ((-0.1, 0.8, 0.05), (120.0, -10.0, 0.0), 110.0)]


@pytest.mark.parametrize("scene_name", ["default", "showcase"])
@pytest.mark.parametrize("W,H", [(1920, 1080), (208, 120)])
def test_camera_rects_never_exclude_an_accepted_hit(native, scene_name, W, H):
    """The screen rectangles of the packet kernel's camera rounds (rvpt_rect.h) are a SUPERSET test: over every pixel x 3 jittered camera rays x every
    triangle, a pair the kernels' float test accepts (interval wide open) never lies outside the triangle's rectangle — camera under / in front of /
    oblique to / inside the model; and the device's rectangles are the host function's (rvpt_camera_rects), word for word."""
    from rvpt_amd import Camera, RenderSettings
    tris, mats, _ = scene_by_name(scene_name)
    ctx = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BRUTE, lab=True)  # the selftests live in the laboratory build (include/rvpt_hip_lab.h)
    try:
        ctx.upload_scene(None, tris, mats)
        seen = 0
        for tr, rot, fov in RECT_CAMERAS:
            c = Camera(W / H)
            c.translation, c.rotation, c.fov = np.array(tr, float), np.array(rot, float), fov
            cam = c.get_data()
            ctx.set_frame(RenderSettings(aa=1, current_frame=5).pack(), cam)
            (accepted, outside, held, pairs), prep, rects = ctx.selftest_camera_rects(3, tris.shape[0])
            assert outside == 0, (tr, accepted, outside)
            assert pairs == tris.shape[0] * ((W + 15) // 16) * ((H + 3) // 4) and held <= pairs
            assert np.array_equal(rects, native.camera_rects(prep, cam, W, H))
            seen += accepted
        assert seen > 0
    finally:
        ctx.close()


def test_packet_kernel_with_and_without_the_rectangles(native, monkeypatch):
    """RVPT_HIP_PACKETS_CULL=0 (no rectangles) and / or RVPT_HIP_PACKETS_BOUNCE_CULL=0 (no bounce cull) against the default: same image, same segment counts — a camera that MOVES between launches in flight (every
    slot's rectangles are rebuilt for the camera of its launch), frames one by one and in batches, a scene swap, a 3-way tile partition, partial edge tiles."""
    from rvpt_amd import Camera, RenderSettings
    W, H = 208, 120

    def run(world=1, rank=0):
        ctx = native.Context(W, H, 0, rank, world, native.TRAVERSAL_BRUTE | native.COUNT_SEGMENTS)
        out = []
        try:
            for scene_name in ("default", "showcase", "default"):
                tris, mats, _ = scene_by_name(scene_name)
                ctx.upload_scene(None, tris, mats)
                for k, (tr, rot, fov) in enumerate(RECT_CAMERAS + RECT_CAMERAS[:2]):
                    c = Camera(W / H)
                    c.translation, c.rotation, c.fov = np.array(tr, float), np.array(rot, float), fov
                    for first, n in ((0, 1), (1, 1), (2, 3)) if k % 2 == 0 else ((0, 5),):
                        ctx.set_frame(RenderSettings(aa=2, current_frame=first).pack(), c.get_data())
                        ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
                    if k % 3 == 2:  # some cameras are read back, the others are overtaken by the next camera's launches while still in flight
                        out.append(ctx.read())
                out.append(ctx.read())
            assert ctx.launch_info()[2] == 6
            return out, ctx.stats()
        finally:
            ctx.close()

    for world, rank in ((1, 0), (3, 1)):
        monkeypatch.delenv("RVPT_HIP_PACKETS_CULL", raising=False)
        monkeypatch.delenv("RVPT_HIP_PACKETS_BOUNCE_CULL", raising=False)
        with_culls, st1 = run(world, rank)
        for rects, bounce in (("0", "1"), ("1", "0"), ("0", "0")):
            monkeypatch.setenv("RVPT_HIP_PACKETS_CULL", rects)
            monkeypatch.setenv("RVPT_HIP_PACKETS_BOUNCE_CULL", bounce)
            without, st0 = run(world, rank)
            assert tuple(st1) == tuple(st0)
            for a, b in zip(with_culls, without):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (rects, bounce)
        assert any(a.any() for a in with_culls)


@pytest.mark.parametrize("W,H,world,rank", [(1920, 1080, 1, 0), (1000, 700, 1, 0), (1920, 1080, 8, 3), (208, 120, 1, 0)])
def test_packet_kernel_claim_order_never_shows_in_the_image(native, monkeypatch, W, H, world, rank):
    """The packet kernel's launches of fewer than four frames deal a frame's 16 x 4 blocks from all over the frame (round 6: FrameParams::perm_*, rvpt_hip_get_cull_info
    bit 5); RVPT_HIP_PACKETS_INTERLEAVE=0 keeps the tile-linear order, -g uses groups of g blocks for launches of every size: the same image and the same segment
    counts, frames one by one and in batches, one and two samples per pixel, a rank's share of eight, an image with partial edge tiles."""
    from rvpt_amd import Camera, RenderSettings
    tris, mats, _ = scene_by_name("default")
    c = Camera(W / H)
    c.translation, c.rotation, c.fov = np.array((0, 0.9, -2.5), float), np.array((0, 0, 0), float), 90.0

    def run():
        ctx = native.Context(W, H, 0, rank, world, native.TRAVERSAL_BRUTE | native.COUNT_SEGMENTS)
        try:
            ctx.upload_scene(None, tris, mats)
            info = []
            for aa in (1, 2):
                for first, n in ((0, 1), (1, 1), (2, 3), (5, 6)):
                    ctx.set_frame(RenderSettings(aa=aa, current_frame=first).pack(), c.get_data())
                    ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
                    info.append(ctx.cull_info())
            ctx.wait()
            assert ctx.launch_info()[2] == 6
            return ctx.read(), tuple(ctx.stats()), info
        finally:
            ctx.close()

    monkeypatch.delenv("RVPT_HIP_PACKETS_INTERLEAVE", raising=False)
    img, st, info = run()
    big = True  # (every size here has at least sixteen groups of blocks to deal and a work plan of whole blocks)
    assert [bool(i & 32) for i in info] == [big, big, big, False] * 2  # the default: launches below four frames, frames large enough to have groups to deal
    assert all(i & 64 for i in info)  # ... all of them through the instances without the uncull'd walks (all three culls are on, the work plans are whole blocks)
    assert img.any()
    for knob, expect in (("0", [False] * 8), ("-1", [big] * 8), ("-4", [big] * 8), ("2", [big, big, big, False] * 2)):
        monkeypatch.setenv("RVPT_HIP_PACKETS_INTERLEAVE", knob)
        img2, st2, info2 = run()
        assert [bool(i & 32) for i in info2] == expect, knob
        assert st2 == st and np.array_equal(img.view(np.uint32), img2.view(np.uint32)), knob


@pytest.mark.parametrize("scene_name", ["default", "showcase"])
def test_bounce_cull_never_excludes_an_accepted_hit(native, scene_name):
    """The bounce cull's table (rvpt_packets.hip: bounce_visibility) is a SUPERSET test: full paths from every pixel, every segment against every triangle with
    the interval wide open — a pair the kernels' float test accepts on a segment that leaves a triangle is always in the row of where it leaves from (Lambert,
    mirror, reflecting and refracting glass in the showcase scene); and the table is worth having (well under all of its bits set)."""
    from rvpt_amd import Camera, RenderSettings
    W, H = 416, 240
    tris, mats, _ = scene_by_name(scene_name)
    ctx = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BRUTE, lab=True)
    try:
        ctx.upload_scene(None, tris, mats)
        total = 0
        for tr, rot, fov in RECT_CAMERAS:
            c = Camera(W / H)
            c.translation, c.rotation, c.fov = np.array(tr, float), np.array(rot, float), fov
            ctx.set_frame(RenderSettings(aa=1, current_frame=3).pack(), c.get_data())
            accepted, outside, bits, size, outside_box, box_tests, box_hits, _ = ctx.selftest_bounce_cull(2)
            assert outside == 0 and outside_box == 0, (tr, accepted, outside, outside_box)
            assert 0 < box_hits < 0.6 * box_tests  # the leaf boxes are worth having: a ray comes near well under all of them
            assert size == 2 * tris.shape[0] ** 2 and 0 < bits < 0.8 * size
            total += accepted
        assert total > 0
    finally:
        ctx.close()


def test_fuzz_culls_slice(native, oracle):
    """A slice of tools/fuzz_culls.py (the full >= 5 000-case run is recorded in profiles/r06_fuzz_culls.txt): random brute-force scenes of 1 .. 1024
    triangles — soups over three decades of size, duplicates / coplanar / interpenetrating / zero-area triangles, slivers at the culls' thresholds, a box of
    large triangles around small geometry, all three materials with odd iors — scaled by 2^-20 .. 2^20, translated across the 64-scale guard, cameras inside / on a
    plane / far / looking away, fov 1 .. 179: the device selftests report nothing outside a rectangle or a row, the four cull on / off combinations render the
    same bits and segment counts on the shipped kernels, and the oracle's brute-force variant agrees."""
    import sys
    sys.path.insert(0, str(ROOT / "tools"))
    import fuzz_culls
    failed, packets = [], 0
    for idx in range(60):
        case = fuzz_culls.make_case(606, idx)
        r = fuzz_culls.run_case(case, 6e7)
        packets += int(bool(r["info"] & 8))
        if r["problems"]:
            failed.append((fuzz_culls.describe(case), r["problems"]))
    assert not failed, failed[:3]
    assert packets >= 40  # most cases reach the packet kernel and both of its culls
    for k in ("RVPT_HIP_PACKETS_CULL", "RVPT_HIP_PACKETS_BOUNCE_CULL"):
        os.environ.pop(k, None)


def test_dispatch_frames_host_counter_and_errors(native, oracle):
    from rvpt_amd import RVPT, scene
    tris, mats = scene.default_scene()

    def make():
        r = RVPT(96, 64, traversal="bvh")
        r.add_triangles(tris)
        for m in mats:
            r.add_material(m)
        r.render_settings.aa = 2
        r.initialize()
        return r

    a, b = make(), make()
    for _ in range(10):
        a.update(); a.draw()
    b.update(); b.draw_frames(7)
    b.update(); b.draw_frames(3)
    assert a.render_settings.current_frame == b.render_settings.current_frame == 9
    assert np.array_equal(a.read_frame(), b.read_frame())
    b.scene_camera.translation = np.array([0.0, 0.5, -1.0])  # moving the camera restarts the accumulation (rvpt.cpp:102-111)
    b.update(); b.draw_frames(2)
    assert b.render_settings.current_frame == 1
    with pytest.raises(native.NativeError):
        b.context.dispatch_frames(0)
    with pytest.raises(native.NativeError):
        b.context.dispatch_frames(native.MAX_FRAMES_PER_DISPATCH + 1)
    a.shutdown(); b.shutdown()


def test_wait_for_times_out_and_completes(native):
    """rvpt_hip_wait_for = the reference's fence wait with its timeout (vk_util.cpp:65,94-97)."""
    from rvpt_amd import Camera, RenderSettings
    tris, mats, nodes = scene_by_name("default")
    ctx = native.Context(1920, 1080, 0, 0, 1, native.TRAVERSAL_BRUTE)
    try:
        ctx.upload_scene(None, tris, mats)
        ctx.set_frame(RenderSettings(aa=8, current_frame=0).pack(), Camera(16 / 9).get_data())
        assert ctx.wait_for(5.0)          # nothing dispatched: done at once
        ctx.dispatch_frames(16)           # tens of milliseconds of work
        assert not ctx.wait_for(1e-6)     # still pending after a microsecond
        assert ctx.wait_for(30.0) and not ctx.query()
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# The HIP path against the reference's own compiled shader (tests/golden/ref_spv, see tests/test_ref_spv.py)

import _refspv  # noqa: E402


def _hip_chain(native, sc, cam, kw, W, H, traversal="bvh", frames=4, flags=0, keep=(0, 3)):
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    fl = flags | {"bvh": native.TRAVERSAL_BVH, "brute": native.TRAVERSAL_BRUTE}[traversal]
    m = kw.get("modes", (9, 9, 9, 9))
    ctx = native.Context(W, H, 0, 0, 1, fl)
    out = {}
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        for f in range(frames):
            rs = RenderSettings(max_bounces=kw.get("max_bounces", 8), aa=kw.get("aa", 1), current_frame=f, camera_mode=kw.get("camera_mode", 0),
                                top_left_render_mode=m[0], top_right_render_mode=m[1], bottom_left_render_mode=m[2], bottom_right_render_mode=m[3],
                                split_ratio=kw.get("split", (0.5, 0.5)))
            ctx.set_frame(rs.pack(), cam)
            ctx.dispatch()
            if f in keep:
                out[f] = ctx.read(native.FORMAT_RGBA8_UNORM) if (flags & native.ACCUM_UNORM8) else ctx.read()
    finally:
        ctx.close()
    return out


@pytest.mark.parametrize("stem,mode", _refspv.mode_cases())
def test_hip_equals_compiled_reference_shader(native, stem, mode):
    """BIT-exact: the HIP kernels (BVH traversal, the reference's live intersect path) against the reference's compiled
    compute_pass.comp.spv executed under the build's contraction rule — all eleven integrators, three cameras, three
    poses, two material sets, frames 0 and 3 of an aa=2 accumulation."""
    sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, mode)
    got = _hip_chain(native, sc, cam, kw, W, H)
    for f in (0, 3):
        assert not got[f][..., 3].any()
        a, b = np.ascontiguousarray(got[f][..., :3]), frames[f]["c"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            f"{stem} mode {mode} frame {f}: {int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())} pixels differ from the reference shader"
        # the uncontracted execution of the same binary: north_star's 1e-4 relative L2 does not hold per image for a
        # chaotic integrator (a 1-ulp change flips a hit), so it is reported through the mean instead
        assert abs(float(a.mean()) - float(frames[f]["u"].mean())) <= 0.02 * max(float(frames[f]["u"].mean()), 1e-3)


def test_hip_brute_force_against_the_reference_shader(native):
    """The LDS-staged brute-force kernel (north_star's deliverable; a closest-hit variant the reference does not have)
    lands on the reference shader's pixels except at exact-t ties / non-conservative slab culls."""
    total = differ = 0
    for stem in ("default_bench_cam0", "default_default_cam0", "default_oblique_cam0", "showcase_bench_cam0", "showcase_oblique_cam0"):
        sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, 9)
        got = _hip_chain(native, sc, cam, kw, W, H, traversal="brute")
        d = (np.ascontiguousarray(got[3][..., :3]).view(np.uint32) != frames[3]["c"].view(np.uint32)).any(axis=2)
        total += d.size
        differ += int(d.sum())
        assert rel_l2(got[3][..., :3], frames[3]["c"]) <= 0.05
    assert differ <= 0.002 * total, (differ, total)


def test_hip_larger_image_and_deeper_tree_against_the_reference_shader(native):
    z = np.load(_refspv.REF / "large_default_bench.npz")
    got = _hip_chain(native, _refspv.load_scene("default"), z["camera"], dict(max_bounces=8, aa=1), 256, 128, frames=1, keep=(0,))
    assert np.array_equal(np.ascontiguousarray(got[0][..., :3]).view(np.uint32), z["f0_c"].view(np.uint32))
    z = np.load(_refspv.REF / "terrain24_kajiya.npz")
    got = _hip_chain(native, _refspv.load_scene("terrain24"), z["camera"], dict(max_bounces=8, aa=2), 64, 32)
    for f in (0, 3):
        assert np.array_equal(np.ascontiguousarray(got[f][..., :3]).view(np.uint32), z[f"f{f}_c"].view(np.uint32)), f"terrain frame {f}"


@pytest.mark.parametrize("name", _refspv.BIG_CASES)
def test_hip_equals_reference_binary_on_large_configurations(native, name):
    """BIT-exact against the reference binary on the shapes of BASELINE C3 / C4 / C5: the 1 002 528-triangle terrain (tree height
    beyond the stack levels kept in LDS: overflow levels, packed stack heads), the Cornell box + 9 152-triangle model, and a
    16 spp x 8-frame accumulation chain."""
    sc, cam, kw, W, H, n_frames, frames = _refspv.load_big_case(name)
    got = _hip_chain(native, sc, cam, kw, W, H, frames=n_frames, keep=tuple(frames))
    for f in frames:
        assert not got[f][..., 3].any()
        a, b = np.ascontiguousarray(got[f][..., :3]), frames[f]["c"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), \
            f"{name} frame {f}: {int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())} pixels differ from the reference shader"


@pytest.mark.parametrize("traversal", ["bvh", "brute"])
def test_hip_converged_mean_agrees_with_a_different_admissible_execution(native, traversal):
    """The reference binary under ANOTHER admissible driver (libm sin/cos/tan, no contraction, IEEE quotient, plain dot / normalize;
    tests/golden/ref_spv/converged_libm.npz) converges to the same image as the HIP kernels: per-pixel z-scores of the 256-frame
    means stay within Monte-Carlo noise (replaces a 2 % single-image tolerance)."""
    z = np.load(_refspv.REF / "converged_libm.npz")
    W, H, N = int(z["width"]), int(z["height"]), int(z["frames"])
    got = _hip_chain(native, _refspv.load_scene("default"), z["camera"], dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"])), W, H,
                     traversal=traversal, frames=N, keep=(N - 1,))[N - 1][..., :3]
    _refspv.assert_converged_agreement(np.ascontiguousarray(got), z)


def test_hip_split_screen_bounce_budget_and_rgba8_against_the_reference_shader(native):
    z = np.load(_refspv.REF / "split_showcase_bench.npz")
    sc = _refspv.load_scene("showcase")
    kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]), modes=tuple(int(m) for m in z["modes"]), split=tuple(float(s) for s in z["split"]))
    got = _hip_chain(native, sc, z["camera"], kw, 64, 32)
    for f in (0, 3):
        assert np.array_equal(np.ascontiguousarray(got[f][..., :3]).view(np.uint32), z[f"f{f}_c"].view(np.uint32)), f"split screen frame {f}"
    z = np.load(_refspv.REF / "bounces2_showcase_bench.npz")
    got = _hip_chain(native, sc, z["camera"], dict(max_bounces=2, aa=1), 64, 32)
    for f in (0, 3):
        assert np.array_equal(np.ascontiguousarray(got[f][..., :3]).view(np.uint32), z[f"f{f}_c"].view(np.uint32)), f"2 bounces frame {f}"
    z = np.load(_refspv.REF / "unorm8_default_bench.npz")
    got = _hip_chain(native, _refspv.load_scene("default"), z["camera"], dict(max_bounces=8, aa=1), 64, 32, frames=6, flags=native.ACCUM_UNORM8, keep=(0, 1, 5))
    for f in (0, 1, 5):
        assert np.array_equal(got[f], z[f"q{f}_c"]), f"rgba8 chain frame {f}"


def test_fast_division_model(native, oracle):
    """The ray/plane quotient of the intersect loop is Markstein's sequence on v_rcp_f32.  Its CPU model
    (oracle o_div_dots == spv_shim.h shim_fdiv_dots) rests on two hardware facts, both checked here on the device:
    the refined reciprocal is the correctly rounded 1/b for EVERY b in [2^-126, 2^126], and v_rcp_f32 flushes outside."""
    mism = native.selftest_rcp()
    assert not mism[1:253].any(), {e: int(m) for e, m in enumerate(mism) if m and e < 253}
    assert int(mism[253]) == 2 ** 23 - 1 and int(mism[254]) == 2 ** 23  # only 2^126 itself has a normal reciprocal up there
    rng = np.random.RandomState(5)
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x807fffff, 0x00400000, 0x00800000, 0x00800001, 0x7e800000, 0x7e800001, 0xfec00000,
                        0x7f000000, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x3f800000, 0x40400000, 0xbf800001, 0x00ffffff, 0x7e7fffff,
                        0x34000000, 0x4b800000, 0x3f7fffff, 0x3fffffff], dtype=np.uint32).view(np.float32)
    a, b = np.meshgrid(special, special)
    # random operands: moderate exponents, full range, and mantissa extremes in the divisor
    n = 200000
    ra = (rng.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)).view(np.float32)
    rb = (rng.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)).view(np.float32)
    mod_a = ((rng.randint(0, 2 ** 23, n) | (rng.randint(100, 155, n) << 23) | (rng.randint(0, 2, n) << 31)).astype(np.uint32)).view(np.float32)
    mod_b = ((rng.choice([0, 1, 0x7fffff, 0x7ffffe, 0x400000], n) | (rng.randint(100, 155, n) << 23)).astype(np.uint32)).view(np.float32)
    aa = np.concatenate([a.ravel(), ra, mod_a])
    bb = np.concatenate([b.ravel(), rb, mod_b])
    got = native.selftest_div(aa, bb)
    want = oracle.div_dots(aa, bb)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
    # on Vulkan's specified domain (divisor in [2^-126, 2^126], nothing leaving the normal range) it IS the IEEE quotient
    with np.errstate(all="ignore"):
        ieee = (mod_a.astype(np.float64) / mod_b.astype(np.float64)).astype(np.float32)
    assert np.array_equal(native.selftest_div(mod_a, mod_b).view(np.uint32), ieee.view(np.uint32))


def test_division_free_pretest_never_stops_what_the_quotient_accepts(native):
    """Camera rounds of the packet kernel decide `0 < t < closest` without the quotient: `!(a > closest * den)` on the sign-normalised
    camera record (rvpt_early_out.h; DESIGN.md 5.1 has the proof sketch).  It must be a SUPERSET test — whatever the quotient accepts
    goes through — for every input: checked on the device on boundary lattices (numerators within a few ulps of closest * den, where the
    two could disagree), +-0, subnormal, huge, inf and NaN operands, closest = inf, records marked not safe, and a random sweep."""
    rng = np.random.RandomState(17)
    f32 = np.float32

    def ulps(x, k):
        return (x.view(np.int32) + k).view(np.float32)

    # (1) the boundary: for random positive den and closest in moderate ranges, a = RN(closest * den) +- {0..3} ulps, and the same around
    #     RN(t * den) for quotient results t just below / at / above closest
    n = 400000
    den = (rng.randint(0, 2 ** 23, n) | (rng.randint(90, 165, n) << 23)).astype(np.uint32).view(np.float32)
    closest = (rng.randint(0, 2 ** 23, n) | (rng.randint(100, 150, n) << 23)).astype(np.uint32).view(np.float32)
    with np.errstate(all="ignore"):
        prod = (closest * den).astype(f32)
    A, D, C = [], [], []
    for k in range(-3, 4):
        A.append(ulps(prod, k)); D.append(den); C.append(closest)
    # (2) special operands, all pairs / triples
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007fffff, 0x00800000, 0x00800001, 0x7e800000, 0x7f000000, 0x7f7fffff, 0x7f800000,
                        0xff800000, 0x7fc00000, 0x3f800000, 0xbf800000, 0x40400000, 0x3f7fffff, 0x3f800001, 0x1e000000, 0x21800000, 0x5d800000, 0x5e000000,
                        0x34000000, 0x4b800000, 0x00ffffff, 0x7e7fffff], dtype=np.uint32).view(np.float32)
    a3, d3, c3 = np.meshgrid(np.abs(special), special, special, indexing="ij")
    A.append(a3.ravel()); D.append(d3.ravel()); C.append(c3.ravel())
    # (3) random bit patterns everywhere (numerator made non-negative: the record holds |num|)
    m = 600000
    ra = rng.randint(0, 2 ** 31, m, dtype=np.int64).astype(np.uint32).view(np.float32)
    rd = rng.randint(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32).view(np.float32)
    rc = rng.randint(0, 2 ** 32, m, dtype=np.uint64).astype(np.uint32).view(np.float32)
    A.append(ra); D.append(rd); C.append(np.where(rng.rand(m) < 0.2, f32(np.inf), rc).astype(f32))
    # (4) scene-like magnitudes, closest = inf and finite
    sa = np.abs(rng.standard_normal(m).astype(f32) * f32(3.0)) + f32(1e-6)
    sd = rng.standard_normal(m).astype(f32)
    sc = np.where(rng.rand(m) < 0.5, f32(np.inf), np.abs(rng.standard_normal(m).astype(f32) * f32(5.0)))
    A.append(sa); D.append(sd); C.append(sc.astype(f32))
    a = np.concatenate(A).astype(f32); d = np.concatenate(D).astype(f32); c = np.concatenate(C).astype(f32)
    bits = native.selftest_pretest(a, d, c)
    through, quotient = (bits & 1) != 0, (bits & 2) != 0
    bad = quotient & ~through
    assert not bad.any(), [(float(a[i]), float(d[i]), float(c[i])) for i in np.flatnonzero(bad)[:8]]
    assert quotient.sum() > 100000 and (~through).sum() > 100000  # the sweep exercises both outcomes
    # and the pre-test is not vacuous: on scene-like operands it stops most of what the quotient rejects
    sl = slice(a.size - m, a.size)
    assert (~through[sl]).sum() > 0.5 * (~quotient[sl]).sum()


# ---------------------------------------------------------------------------------------------------------------------
# The RCCL gather inside the C ABI (rvpt_hip_comm_* / rvpt_hip_gather / collective rvpt_hip_read).  One GPU here, so the
# communicator has one rank and the gather is a self send/recv through RCCL — the same calls the N-rank case makes.

def _render_some(native, ctx, sc, cam, frames=3, aa=2):
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    ctx.upload_scene(None, tris, mats)
    for f in range(frames):
        ctx.set_frame(RenderSettings(aa=aa, current_frame=f).pack(), cam)
        ctx.dispatch()


def test_collective_read_through_the_library_communicator(native):
    import torch
    sc = scene_by_name("default")
    W, H = 100, 52  # partial edge tiles
    cam = identity_camera(W / H)
    plain = native.Context(W, H, 0, 0, 1, 0)
    coll = native.Context(W, H, 0, 0, 1, 0)
    try:
        _render_some(native, plain, sc, cam)
        _render_some(native, coll, sc, cam)
        want, want8 = plain.read(), plain.read(native.FORMAT_RGBA8_UNORM)
        coll.comm_init(native.comm_unique_id())
        with pytest.raises(native.NativeError, match="already has a communicator"):
            coll.comm_init(native.comm_unique_id())
        ranks, rank, version = coll.comm_info()                               # asked of RCCL: ncclCommCount / ncclCommUserRank / ncclGetVersion
        assert (ranks, rank) == (1, 0) and version >= 20000
        with pytest.raises(native.NativeError, match="no communicator"):
            plain.comm_info()
        assert np.array_equal(coll.read(), want)                              # gather -> untile -> host
        assert np.array_equal(coll.read(native.FORMAT_RGBA8_UNORM), want8)
        out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        coll.gather(out.data_ptr())                                           # the same, left on the device
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        # rendering goes on after a gather: the accumulator was only read
        from rvpt_amd import RenderSettings
        for c in (plain, coll):
            c.set_frame(RenderSettings(aa=2, current_frame=3).pack(), cam)
            c.dispatch()
        assert np.array_equal(coll.read(), plain.read())
    finally:
        plain.close()
        coll.close()


def test_single_process_group_and_its_errors(native):
    sc = scene_by_name("default")
    W, H = 64, 48
    cam = identity_camera(W / H)
    a = native.Context(W, H, 0, 0, 1, 0)
    b = native.Context(W, H, 0, 1, 2, 0)
    try:
        with pytest.raises(native.NativeError, match="must list ranks"):
            native.comm_init_all([b])          # rank 1 of 2 alone is not a group
        _render_some(native, a, sc, cam)
        want = a.read()
        native.comm_init_all([a])              # ncclCommInitAll over this process's contexts (here: one)
        assert np.array_equal(a.read(), want)
        with pytest.raises(native.NativeError, match="no communicator"):
            b.gather(None)
    finally:
        a.close()
        b.close()


def test_distributed_wrapper_uses_the_library_collective(native, monkeypatch):
    """DistributedRVPT with a forced world-1 collective: the communicator id travels, rvpt_hip_gather does the rest."""
    from rvpt_amd import scene
    from rvpt_amd.distributed import DistributedRVPT
    monkeypatch.setenv("RVPT_FORCE_COLLECTIVE", "1")
    tris, mats = scene.default_scene()
    imgs = []
    for forced in (True, False):
        if not forced:
            monkeypatch.delenv("RVPT_FORCE_COLLECTIVE")
        r = DistributedRVPT(96, 64, traversal="brute", rank=0, world=1, device=0)
        r.add_triangles(tris)
        for m in mats:
            r.add_material(m)
        assert r.initialize()
        assert r.library_comm == forced
        for _ in range(3):
            r.update()
            r.draw()
        imgs.append(r.read_frame())
        r.shutdown()
    assert np.array_equal(imgs[0], imgs[1])


def _two_rank_worker(rank, world, port, W, H, out_path, shared_gpu=False, traversal="brute", fail_comm_rank=None):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = 0 if shared_gpu else rank
    torch.cuda.set_device(device)
    # torch.distributed is the control plane only (gloo); the process's one RCCL communicator is the library's.  Ranks sharing the
    # one GPU of a test box cannot form one (RCCL refuses two ranks on a device): no library communicator, gather staged through the host
    if shared_gpu and fail_comm_rank is None:
        os.environ["RVPT_NO_LIBRARY_COMM"] = "1"
    if fail_comm_rank is not None:  # one rank cannot join: every rank must fall back together — a clean message, not a hang
        os.environ["RVPT_TEST_FAIL_COMM_RANK"] = str(fail_comm_rank)
        os.environ["RVPT_HIP_COMM_TIMEOUT_S"] = "20"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rvpt_amd import scene
        from rvpt_amd.distributed import DistributedRVPT
        tris, mats = scene.default_scene()
        r = DistributedRVPT(W, H, traversal=traversal, rank=rank, world=world, device=device)
        r.add_triangles(tris)
        for m in mats:
            r.add_material(m)
        assert r.initialize()
        assert r.library_comm == (not shared_gpu)  # real devices: the gather runs inside the C ABI
        r.barrier()
        for _ in range(3):
            r.update()
            r.draw()
        r.barrier()
        img = r.read_frame()
        if rank == 0:
            np.save(out_path, img)
        dist.barrier()
        r.shutdown()
    finally:
        dist.destroy_process_group()


def test_two_processes_two_gpus_gather_through_the_library(native, tmp_path):
    """One process per GPU, the tile split and the RCCL gather of rvpt_hip_gather across two real devices == one GPU.
    Needs two GPUs (skipped on the one-GPU test boxes; the driver's scaling run is the other place this path executes)."""
    if native.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    from rvpt_amd import RVPT, scene
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H = 208, 112
    out = tmp_path / "img.npy"
    mp.spawn(_two_rank_worker, args=(2, port, W, H, str(out)), nprocs=2, join=True)
    tris, mats = scene.default_scene()
    r = RVPT(W, H, traversal="brute")
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.initialize()
    for _ in range(3):
        r.update()
        r.draw()
    want = r.read_frame()
    r.shutdown()
    assert np.array_equal(np.load(out), want)


def test_a_rank_that_cannot_join_the_communicator_makes_every_rank_fall_back(native, tmp_path):
    """Rank 1 fails before the RCCL bootstrap (injected): the ranks agree over the control plane BEFORE any of them blocks in
    ncclCommInitRank, every rank falls back to the host-staged gather with a message, and the frame is still right — no hang."""
    import socket
    import torch.multiprocessing as mp
    from rvpt_amd import RVPT, scene
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H = 208, 112
    out = tmp_path / "img.npy"
    mp.spawn(_two_rank_worker, args=(2, port, W, H, str(out), True, "brute", 1), nprocs=2, join=True)
    tris, mats = scene.default_scene()
    r = RVPT(W, H, traversal="brute")
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.initialize()
    for _ in range(3):
        r.update()
        r.draw()
    want = r.read_frame()
    r.shutdown()
    assert np.array_equal(np.load(out), want)


def test_ranks_that_rccl_refuses_fall_back_cleanly(native, tmp_path):
    """Two processes on the ONE GPU of a test box try for the library communicator in earnest (no RVPT_NO_LIBRARY_COMM): RCCL
    refuses two ranks on one device — a real ncclCommInitRank failure (or, at worst, a bootstrap that the 20 s deadline ends), on
    every rank — and the ranks agree to leave it, drop what they joined, and gather through the host.  The frame is still right."""
    import socket
    import torch.multiprocessing as mp
    from rvpt_amd import RVPT, scene
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H = 208, 112
    out = tmp_path / "img.npy"
    mp.spawn(_two_rank_worker, args=(2, port, W, H, str(out), True, "brute", -1), nprocs=2, join=True)  # fail_comm_rank = -1: nobody is made to fail
    tris, mats = scene.default_scene()
    r = RVPT(W, H, traversal="brute")
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.initialize()
    for _ in range(3):
        r.update()
        r.draw()
    want = r.read_frame()
    r.shutdown()
    assert np.array_equal(np.load(out), want)


def test_collective_errors_are_reported_not_hung(native):
    """Rank 0's own bad arguments in a collective read are reported AFTER it has taken part in the exchange; a barrier and a
    gather on a context without a communicator say so; comm_destroy makes reads local again."""
    sc = scene_by_name("default")
    W, H = 64, 48
    cam = identity_camera(W / H)
    c = native.Context(W, H, 0, 0, 1, 0)
    try:
        _render_some(native, c, sc, cam)
        want = c.read()
        with pytest.raises(native.NativeError, match="no communicator"):
            c.comm_barrier()
        c.comm_init(native.comm_unique_id())
        c.comm_barrier()
        small = np.zeros(16, np.float32)
        rc = native.load().rvpt_hip_read(c._h, native.FORMAT_RGBA32F, small.ctypes.data, small.nbytes)
        assert rc == native.ERR_SIZE and b"frame needs" in native.load().rvpt_hip_last_error(c._h)
        rc = native.load().rvpt_hip_read(c._h, 77, small.ctypes.data, 1 << 30)
        assert rc == native.ERR_INVALID
        with pytest.raises(native.NativeError, match="needs a destination"):
            c.gather(None)
        assert np.array_equal(c.read(), want)  # the communicator survived all of it
        c.comm_barrier()
        c.comm_destroy()
        with pytest.raises(native.NativeError, match="no communicator"):
            c.gather(None)
        assert np.array_equal(c.read(), want)  # a local read again
    finally:
        c.close()


@pytest.mark.parametrize("world,traversal", [(2, "brute"), (3, "bvh"), (8, "brute")])
def test_processes_sharing_one_gpu_partition_and_gather(native, tmp_path, world, traversal):
    """The N-process flow on the ONE GPU a test box has: every rank is its own process with its own context, tile share and
    accumulator (all on cuda:0), the process group is gloo and the frame gather is staged through the host — everything of the
    multi-GPU path except RCCL's transport.  The gathered frame == the unsplit frame, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    from rvpt_amd import RVPT, scene
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H = 208, 112
    out = tmp_path / "img.npy"
    mp.spawn(_two_rank_worker, args=(world, port, W, H, str(out), True, traversal), nprocs=world, join=True)
    tris, mats = scene.default_scene()
    r = RVPT(W, H, traversal=traversal)
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.initialize()
    for _ in range(3):
        r.update()
        r.draw()
    want = r.read_frame()
    r.shutdown()
    assert np.array_equal(np.load(out), want)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_under_torchrun_with_ranks_sharing_one_gpu(native, world):
    """bench.py exactly as the driver launches it for N = 2 (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2),
    with RVPT_BENCH_SHARED_GPU=1 so that both ranks run on the box's one GPU: barriers, the MAX over ranks, the SUM of the
    statistics, rank 0's single JSON line as the LAST line of stdout."""
    import json
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RVPT_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "20" if world == 8 else "6", "--warmup", "5" if world == 8 else "2", "--width", "640", "--height", "360",
           "--no-cpu-baseline", "--ramp-seconds", "0"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["steps"] == (20 if world == 8 else 6) and line["value"] > 0 and line["scaling"] == "strong"
    assert line["config"]["parallelism"].startswith(f"tile{world}")
    if world == 8:
        assert line["config"]["launches"] == [20]  # the driver's 20 steps at batch 64: one launch (measured best on a small tile share)
    assert abs(line["config"]["segments_per_sample"] - 1.44) < 0.05  # both ranks' statistics were summed
    # the collective describes itself: ranks sharing one GPU cannot form an RCCL communicator, and the line says so instead of passing for RCCL
    assert line["collective"]["path"] == "host-staged" and line["collective"]["rccl_ranks"] == 0 and line["collective"]["gather_ms"] > 0
    assert line["value_one_frame_per_launch"]["value"] > 0 and line["value_one_frame_per_launch"]["frames"] >= 20


def test_bench_with_gpus_n_fans_out_by_itself_or_fails(native):
    """`python bench.py --gpus 8` WITHOUT a launcher: it must either run eight ranks — here all on the box's one GPU,
    RVPT_BENCH_SHARED_GPU=1 — and print one JSON line with n_gpus 8, or fail with the reason when fewer than eight devices are
    visible.  Never a one-GPU run labelled as N."""
    import json
    import subprocess
    import sys
    import torch
    base = [sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--width", "640", "--height", "360", "--no-cpu-baseline",
            "--ramp-seconds", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RVPT_BENCH_SHARED_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if torch.cuda.device_count() < 8:
        res = subprocess.run(base, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode != 0
        assert f"8 GPUs requested, {torch.cuda.device_count()} visible" in res.stderr
        assert "n_gpus" not in res.stdout
    res = subprocess.run(base, env=dict(env, RVPT_BENCH_SHARED_GPU="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["steps"] == 20 and line["value"] > 0
    assert line["config"]["parallelism"].startswith("tile8")
    assert line["collective"]["path"] == "host-staged" and "rccl_version" in line["collective"]


def test_render_cli_refuses_more_gpus_than_visible(native):
    """rvpt_render --gpus N with fewer than N devices: a one-line reason and a non-zero exit, not a run on fewer."""
    import subprocess
    import torch
    from rvpt_amd import build as rv_build
    exe = rv_build.build_host() / "rvpt_render"
    n = torch.cuda.device_count() + 1
    res = subprocess.run([str(exe), "--obj", str(ROOT / "does_not_matter.obj"), "--gpus", str(n)], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and f"{n} GPUs requested, {n - 1} visible" in res.stderr


def test_growing_the_sample_buffers_does_not_race_with_the_launch(native):
    """The first dispatch_frames(n) of a context grows its per-launch sample buffers and zeroes them; the zeroing must be
    ordered before the frame kernel on the (non-blocking) slot stream.  Found by tools/fuzz_parity.py (1 case in 1 500);
    repeated here on fresh contexts, partitioned image, small frames, where the window is widest."""
    from rvpt_amd import RenderSettings
    sc = scene_by_name("showcase")
    tris, mats, nodes = sc
    W, H = 89, 43
    cam = identity_camera(W / H)

    def render(plan, rank):
        ctx = native.Context(W, H, 0, rank, 3, native.TRAVERSAL_BRUTE)
        try:
            ctx.upload_scene(None, tris, mats)
            f = 0
            for n in plan:
                ctx.set_frame(RenderSettings(max_bounces=2, aa=1, current_frame=f).pack(), cam)
                ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
                f += n
            return ctx.read()
        finally:
            ctx.close()

    want = [render([1, 1, 1, 1], r) for r in range(3)]
    for rep in range(60):
        for r, plan in enumerate(([3, 1], [1, 1, 2], [2, 1, 1])):
            got = render(plan, r)
            assert np.array_equal(got.view(np.uint32), want[r].view(np.uint32)), f"repetition {rep}, rank {r}, plan {plan}"
