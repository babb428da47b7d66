"""BASELINE.json's configurations on the GPU, at their full sizes through size-independent properties and at sizes the
oracle covers through bit comparison.

  C2  Cornell box + 9 152-triangle model, 1920x1080, 4 spp                      (intersect-loop stress)
  C3  ~1 M-triangle heightfield, 1920x1080, 1 spp, tile-split over 8 ranks
  C4  3840x2160, 16 spp, 64-frame temporal accumulation = 1 024 samples per pixel, 8 ranks
(C0 / C1, the default scene at 256x256 and 1920x1080, are in tests/test_gpu_parity.py.)

Properties asserted at full size, all bit-exact:
  (a) the regenerating (persistent, ballot-compacted) kernel == the one-pixel-per-lane kernel;
  (b) the tiles of every rank of an N-way partition (N = 2, 3, 8; emulated one rank at a time on this GPU) are exactly the
      unsplit frame's pixels, every pixel owned once;
  (c) n accumulation frames issued as one launch (rvpt_hip_dispatch_frames) == n launches.
"""
import numpy as np
import pytest

from _util import scene_by_name

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from rvpt_amd import build, native as n
    build.build_native()
    n.load()
    assert n.device_count() >= 1
    return n


def render(native, sc, cam, W, H, traversal, plan, aa, flags=0, world=1, rank=0, max_bounces=8):
    """plan: [(first_frame, n_frames), ...] — each entry one dispatch (n_frames > 1: rvpt_hip_dispatch_frames)."""
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    fl = flags | {"bvh": native.TRAVERSAL_BVH, "brute": native.TRAVERSAL_BRUTE}[traversal]
    ctx = native.Context(W, H, 0, rank, world, fl)
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        for first, n in plan:
            ctx.set_frame(RenderSettings(aa=aa, current_frame=first, max_bounces=max_bounces).pack(), cam)
            if n == 1:
                ctx.dispatch()
            else:
                ctx.dispatch_frames(n)
        return ctx.read()
    finally:
        ctx.close()


def same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def check_partitions(native, full, sc, cam, W, H, traversal, plan, aa, worlds):
    tiles_x = (W + 15) // 16
    ty, tx = np.meshgrid(np.arange(H) // 16, np.arange(W) // 16, indexing="ij")
    from rvpt_amd.distributed import tile_slot
    tile = tile_slot(tx, ty, tiles_x)
    for world in worlds:
        covered = np.zeros((H, W), dtype=bool)
        for rank in range(world):
            part = render(native, sc, cam, W, H, traversal, plan, aa, world=world, rank=rank)
            mine = (tile % world) == rank
            assert not part[~mine].any(), f"{world}-way: rank {rank} wrote a foreign tile"
            assert same(part[mine], full[mine]), f"{world}-way: rank {rank}'s tiles differ from the unsplit frame"
            assert not (covered & mine).any()
            covered |= mine
        assert covered.all()


def cornell_camera(aspect):
    from rvpt_amd import Camera
    c = Camera(aspect)
    c.translation = np.array([0.0, 2.0, -1.9])
    return c.get_data()


def heightfield_camera(aspect):
    from rvpt_amd import Camera
    c = Camera(aspect)
    c.translation = np.array([0.0, 2.5, -5.0])
    c.rotation = np.array([0.0, 25.0, 0.0])
    return c.get_data()


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle as o
    o.build()
    return o


def test_config4_accumulation_to_1024_samples_vs_oracle(native, oracle_mod):
    """C4's arithmetic at a size the oracle covers: Cornell geometry, 16 spp, current_frame 0..63 — a 1 024-sample running
    mean — issued as 8 launches of 8 frames, against the oracle's frame-by-frame chain.  Bit-exact, BVH (the reference's
    traversal); the LDS-streamed brute-force kernel over a shorter chain."""
    o = oracle_mod
    sc = scene_by_name("cornell")
    tris, mats, nodes = sc
    W, H = 64, 36
    cam = cornell_camera(W / H)
    got = render(native, sc, cam, W, H, "bvh", [(f, 8) for f in range(0, 64, 8)], aa=16)
    prev = None
    for f in range(64):
        prev, _ = o.render(o.settings_bytes(aa=16, current_frame=f), cam, nodes, tris, mats, W, H, o.TRAVERSAL_BVH, prev=prev)
    assert same(got, prev), f"{int((got != prev).any(axis=2).sum())} of {W * H} pixels differ after 64 frames x 16 spp"
    assert float(got[..., :3].mean()) > 0.05 and np.isfinite(got).all()  # the light is found: a converging image, not black
    W, H = 32, 18
    cam = cornell_camera(W / H)
    got = render(native, sc, cam, W, H, "brute", [(0, 4), (4, 4)], aa=16)
    prev = None
    for f in range(8):
        prev, _ = o.render(o.settings_bytes(aa=16, current_frame=f), cam, nodes, tris, mats, W, H, o.TRAVERSAL_BRUTE, prev=prev)
    assert same(got, prev)


def test_config4_full_size_3840x2160_16spp(native):
    sc = scene_by_name("cornell")
    W, H, aa = 3840, 2160, 16
    cam = cornell_camera(W / H)
    plan = [(0, 1), (1, 1)]
    full = render(native, sc, cam, W, H, "bvh", plan, aa)
    assert np.isfinite(full).all() and not full[..., 3].any()
    assert same(render(native, sc, cam, W, H, "bvh", plan, aa, flags=native.KERNEL_SIMPLE), full)      # (a)
    assert same(render(native, sc, cam, W, H, "bvh", [(0, 2)], aa), full)                              # (c)
    check_partitions(native, full, sc, cam, W, H, "bvh", plan, aa, worlds=(2, 3, 8))                   # (b)


@pytest.mark.parametrize("traversal", ["bvh", "brute"])
def test_config2_full_size_1920x1080_4spp_cornell(native, traversal):
    sc = scene_by_name("cornell")
    W, H, aa = 1920, 1080, 4
    cam = cornell_camera(W / H)
    plan = [(0, 1), (1, 1)]
    full = render(native, sc, cam, W, H, traversal, plan, aa)
    assert np.isfinite(full).all()
    assert same(render(native, sc, cam, W, H, traversal, plan, aa, flags=native.KERNEL_SIMPLE), full)
    assert same(render(native, sc, cam, W, H, traversal, [(0, 2)], aa), full)
    check_partitions(native, full, sc, cam, W, H, traversal, plan, aa, worlds=(2, 8) if traversal == "bvh" else (3,))


def test_launches_in_flight_follow_the_launch_and_the_frames_stay_in_order(native, monkeypatch):
    """The HBM-resident BVH kernel rotates short launches (one 1080p x 4 spp frame: 8.3 M samples) over six slots and long ones
    (eight frames) over three; going back and forth needs no drain and must not reorder the temporal blend: any mix of single and
    batched dispatches gives the accumulation of the same frames one by one with the number of slots pinned."""
    sc = scene_by_name("cornell")
    W, H, aa = 1920, 1080, 4
    cam = cornell_camera(W / H)
    mixed = [(0, 1), (1, 1), (2, 8), (10, 1), (11, 1), (12, 1), (13, 1), (14, 1), (15, 1), (16, 1), (17, 8), (25, 8), (33, 1)]
    got = render(native, sc, cam, W, H, "bvh", mixed, aa)
    monkeypatch.setenv("RVPT_HIP_FRAMES_IN_FLIGHT", "2")
    want = render(native, sc, cam, W, H, "bvh", [(f, 1) for f in range(34)], aa)
    assert same(got, want)


def test_config2_brute_force_and_bvh_agree_at_full_size(native):
    """Like-for-like traversals differ only at exact-t ties and non-conservative slab culls."""
    sc = scene_by_name("cornell")
    W, H = 1920, 1080
    cam = cornell_camera(W / H)
    a = render(native, sc, cam, W, H, "bvh", [(0, 1)], 4)
    b = render(native, sc, cam, W, H, "brute", [(0, 1)], 4)
    differing = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
    assert differing <= 2e-3 * W * H, differing


def test_config3_full_size_1920x1080_million_triangles(native):
    from rvpt_amd import scene
    tris, mats = scene.heightfield_scene()
    assert tris.shape[0] == 1002528
    nodes, idx = native.build_bvh(tris)
    sc = (tris[idx], mats, nodes)
    W, H, aa = 1920, 1080, 1
    cam = heightfield_camera(W / H)
    plan = [(0, 1), (1, 1), (2, 1)]
    full = render(native, sc, cam, W, H, "bvh", plan, aa)
    assert np.isfinite(full).all()
    assert same(render(native, sc, cam, W, H, "bvh", plan, aa, flags=native.KERNEL_SIMPLE), full)
    assert same(render(native, sc, cam, W, H, "bvh", [(0, 3)], aa), full)
    check_partitions(native, full, sc, cam, W, H, "bvh", plan, aa, worlds=(8,))
    # the LDS-streamed brute-force loop over the same million triangles, at a size it finishes quickly: same pixels as the
    # BVH but for ties / slab culls
    w, h = 240, 135
    cam = heightfield_camera(w / h)
    a = render(native, sc, cam, w, h, "bvh", [(0, 1)], 1)
    b = render(native, sc, cam, w, h, "brute", [(0, 1)], 1)
    assert int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum()) <= 5e-3 * w * h
