"""GPU parity of the wavefront BVH pipeline (rvpt_amd/csrc/rvpt_wavefront.hip: begin / traverse / shade kernels, path records in HBM)
against the reference binary's fixtures, the CPU oracle and the megakernel it stands in for — bit for bit in every case: the order of
operations per pixel is the megakernel's.  The pipeline is opt-in (RVPT_HIP_BVH_WAVEFRONT: wherever it is eligible); it measured
slower than the megakernel on MI355X, so the default policy never selects it; RVPT_HIP_BVH_MEGAKERNEL rules it out."""
import numpy as np
import pytest

import _refspv
from _util import scene_by_name

pytestmark = pytest.mark.gpu

VARIANT_WAVEFRONT = 4


@pytest.fixture(scope="module")
def native():
    from rvpt_amd import build, native as n
    build.build_native()
    n.load()
    assert n.device_count() >= 1
    return n


def _render(native, sc, cam, W, H, plan, kw, traversal="bvh", flags=0, world=1, rank=0, keep=None, want_variant=None):
    """plan = [(first_frame, n_frames), ...]; returns ({frame index after which it was read: image}, stats, kernel variant of the last launch)."""
    from rvpt_amd import RenderSettings
    tris, mats, nodes = sc
    fl = flags | native.COUNT_SEGMENTS | {"bvh": native.TRAVERSAL_BVH, "bvh_ordered": native.TRAVERSAL_BVH_ORDERED, "brute": native.TRAVERSAL_BRUTE}[traversal]
    ctx = native.Context(W, H, 0, rank, world, fl)
    out = {}
    try:
        ctx.upload_scene(nodes if traversal != "brute" else None, tris, mats)
        for first, n in plan:
            rs = RenderSettings(max_bounces=kw.get("max_bounces", 8), aa=kw.get("aa", 1), current_frame=first)
            ctx.set_frame(rs.pack(), cam)
            ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
            last = first + n - 1
            if keep is None or last in keep:
                out[last] = ctx.read(native.FORMAT_RGBA8_UNORM) if (flags & native.ACCUM_UNORM8) else ctx.read()
        variant = ctx.launch_info()[2]
        if want_variant is not None:
            assert variant == want_variant, f"kernel variant {variant}, expected {want_variant}"
        return out, ctx.stats(), variant
    finally:
        ctx.close()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


KAJIYA_PINHOLE = [stem for stem, mode in _refspv.mode_cases() if mode == 9 and stem.endswith("cam0")]


@pytest.mark.parametrize("stem", KAJIYA_PINHOLE)
def test_wavefront_equals_compiled_reference_shader(native, stem):
    """The wavefront pipeline against the reference's compiled compute_pass.comp.spv (contracted execution), frames 0 and 3 of an
    aa = 2 accumulation: default, showcase (mirror / glass / emitter) and Cornell scenes, three camera poses."""
    sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, 9)
    got, _, _ = _render(native, sc, cam, W, H, [(f, 1) for f in range(4)], kw, flags=native.BVH_WAVEFRONT, keep=(0, 3), want_variant=VARIANT_WAVEFRONT)
    for f in (0, 3):
        assert not got[f][..., 3].any()
        a, b = _bits(got[f][..., :3]), _bits(frames[f]["c"])
        assert np.array_equal(a, b), f"{stem} frame {f}: {int((a != b).any(axis=2).sum())} pixels differ from the reference shader"


@pytest.mark.parametrize("name", _refspv.BIG_CASES)
def test_wavefront_equals_reference_binary_on_large_configurations(native, name):
    """1 M-triangle terrain (stack overflow levels, packed heads), Cornell + model, and the 16 spp x 8-frame chain (per-pixel sample
    sums through HBM, the RNG stream running on from sample to sample) — as one launch per frame and as batched launches."""
    sc, cam, kw, W, H, n_frames, frames = _refspv.load_big_case(name)
    one, _, _ = _render(native, sc, cam, W, H, [(f, 1) for f in range(n_frames)], kw, flags=native.BVH_WAVEFRONT, keep=tuple(frames), want_variant=VARIANT_WAVEFRONT)
    for f in frames:
        assert np.array_equal(_bits(one[f][..., :3]), _bits(frames[f]["c"])), f"{name} frame {f}"
    last = max(frames)
    batched, _, _ = _render(native, sc, cam, W, H, [(0, 3), (3, n_frames - 3)], kw, flags=native.BVH_WAVEFRONT, keep=(last,), want_variant=VARIANT_WAVEFRONT)
    assert np.array_equal(_bits(batched[last][..., :3]), _bits(frames[last]["c"])), f"{name} batched"


def test_wavefront_other_reference_cases(native):
    """Larger image, deeper tree, an exhausted bounce budget, and the reference's rgba8 accumulation (quantised every frame)."""
    z = np.load(_refspv.REF / "large_default_bench.npz")
    got, _, _ = _render(native, _refspv.load_scene("default"), z["camera"], 256, 128, [(0, 1)], dict(max_bounces=8, aa=1), flags=native.BVH_WAVEFRONT, want_variant=VARIANT_WAVEFRONT)
    assert np.array_equal(_bits(got[0][..., :3]), _bits(z["f0_c"]))
    z = np.load(_refspv.REF / "terrain24_kajiya.npz")
    got, _, _ = _render(native, _refspv.load_scene("terrain24"), z["camera"], 64, 32, [(0, 2), (2, 2)], dict(max_bounces=8, aa=2), flags=native.BVH_WAVEFRONT, keep=(3,))
    assert np.array_equal(_bits(got[3][..., :3]), _bits(z["f3_c"]))
    z = np.load(_refspv.REF / "bounces2_showcase_bench.npz")
    got, _, _ = _render(native, _refspv.load_scene("showcase"), z["camera"], 64, 32, [(f, 1) for f in range(4)], dict(max_bounces=2, aa=1), flags=native.BVH_WAVEFRONT, keep=(0, 3))
    for f in (0, 3):
        assert np.array_equal(_bits(got[f][..., :3]), _bits(z[f"f{f}_c"]))
    z = np.load(_refspv.REF / "unorm8_default_bench.npz")
    got, _, _ = _render(native, _refspv.load_scene("default"), z["camera"], 64, 32, [(f, 1) for f in range(6)], dict(max_bounces=8, aa=1),
                        flags=native.BVH_WAVEFRONT | native.ACCUM_UNORM8, keep=(0, 1, 5), want_variant=VARIANT_WAVEFRONT)
    for f in (0, 1, 5):
        assert np.array_equal(got[f], z[f"q{f}_c"]), f"rgba8 chain frame {f}"


@pytest.mark.parametrize("traversal", ["bvh", "bvh_ordered"])
def test_wavefront_equals_megakernel(native, traversal):
    """Same image and same segment / sample counts as the megakernel: partial edge tiles, a 3-way tile partition, batches of frames,
    both visiting orders, 1 bounce."""
    from rvpt_amd import Camera
    W, H = 176, 104  # partial edge tiles
    sc = scene_by_name("cornell")
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()
    for kw, plan, world, rank in ((dict(max_bounces=8, aa=3), [(0, 2), (2, 1), (3, 4)], 1, 0),
                                  (dict(max_bounces=8, aa=1), [(0, 5)], 3, 1),
                                  (dict(max_bounces=1, aa=2), [(0, 1), (1, 2)], 1, 0)):
        last = plan[-1][0] + plan[-1][1] - 1
        mega, st_m, v_m = _render(native, sc, cam, W, H, plan, kw, traversal, flags=native.BVH_MEGAKERNEL, world=world, rank=rank, keep=(last,))
        wave, st_w, v_w = _render(native, sc, cam, W, H, plan, kw, traversal, flags=native.BVH_WAVEFRONT, world=world, rank=rank, keep=(last,))
        assert v_m == 2 and v_w == VARIANT_WAVEFRONT
        assert np.array_equal(_bits(mega[last]), _bits(wave[last])), (kw, plan, world)
        assert tuple(st_m) == tuple(st_w), (st_m, st_w)
        assert wave[last].any() or kw["max_bounces"] == 1  # (one segment inside the box: every path ends on its bounce budget, black)


def test_wavefront_full_hd_and_default_policy(native):
    """1920x1080 x 4 spp on the Cornell scene (BASELINE C3's shape, two frames as one launch): the forced wavefront pipeline equals the
    megakernel; nothing selects the pipeline by default (it measured slower), and a generic render mode never runs it."""
    from rvpt_amd import Camera, RenderSettings
    W, H = 1920, 1080
    sc = scene_by_name("cornell")
    c = Camera(W / H)
    c.translation = np.array([0.0, 2.0, -1.9])
    cam = c.get_data()
    kw = dict(max_bounces=8, aa=4)
    wave, st_w, v_w = _render(native, sc, cam, W, H, [(0, 2)], kw, flags=native.BVH_WAVEFRONT)
    mega, st_m, v_m = _render(native, sc, cam, W, H, [(0, 2)], kw)
    assert v_w == VARIANT_WAVEFRONT and v_m == 2  # default policy: the megakernel
    assert np.array_equal(_bits(wave[1]), _bits(mega[1])) and tuple(st_w) == tuple(st_m)
    tris, mats, nodes = sc
    ctx = native.Context(W, H, 0, 0, 1, native.TRAVERSAL_BVH | native.BVH_WAVEFRONT)
    try:
        ctx.upload_scene(nodes, tris, mats)
        ctx.set_frame(RenderSettings(aa=1, current_frame=0, top_left_render_mode=5).pack(), cam)  # ambient occlusion in one quadrant: generic kernel
        ctx.dispatch()
        ctx.wait()
        assert ctx.launch_info()[2] == 2
    finally:
        ctx.close()


VARIANT_BRUTE_WAVEFRONT = 5


@pytest.mark.parametrize("scene_name", ["default", "showcase"])
def test_brute_wavefront_equals_brute_megakernel(native, monkeypatch, scene_name):
    """The wavefront form of the brute-force path (wf_trace_brute: packets of 64 records of one tile, the camera-ray iteration with the
    packet-uniform early-out on the plane distance) against the resident brute-force megakernel: same image and same segment / sample
    counts — partial edge tiles, 3 spp, batches of frames, a 3-way tile partition, one bounce; with and without the early-out."""
    from rvpt_amd import Camera
    W, H = 176, 104
    sc = scene_by_name(scene_name)
    c = Camera(W / H)
    c.translation = np.array([0.1, 0.9, -2.4])
    cam = c.get_data()
    for kw, plan, world, rank in ((dict(max_bounces=8, aa=3), [(0, 2), (2, 1), (3, 4)], 1, 0),
                                  (dict(max_bounces=8, aa=1), [(0, 5)], 3, 1),
                                  (dict(max_bounces=1, aa=2), [(0, 1), (1, 2)], 1, 0)):
        last = plan[-1][0] + plan[-1][1] - 1
        mega, st_m, v_m = _render(native, sc, cam, W, H, plan, kw, "brute", flags=native.BRUTE_MIXED_PACKETS, world=world, rank=rank, keep=(last,))
        wave, st_w, v_w = _render(native, sc, cam, W, H, plan, kw, "brute", flags=native.BRUTE_WAVEFRONT, world=world, rank=rank, keep=(last,))
        assert v_m == 0 and v_w == VARIANT_BRUTE_WAVEFRONT
        assert np.array_equal(_bits(mega[last]), _bits(wave[last])), (kw, plan, world)
        assert tuple(st_m) == tuple(st_w), (st_m, st_w)
        assert wave[last].any()
    monkeypatch.setenv("RVPT_HIP_WF_NO_EARLY_OUT", "1")
    kw, plan = dict(max_bounces=8, aa=2), [(0, 3)]
    plain, _, v = _render(native, sc, cam, W, H, plan, kw, "brute", flags=native.BRUTE_WAVEFRONT, keep=(2,))
    monkeypatch.delenv("RVPT_HIP_WF_NO_EARLY_OUT")
    early, _, _ = _render(native, sc, cam, W, H, plan, kw, "brute", flags=native.BRUTE_WAVEFRONT, keep=(2,))
    mega, _, _ = _render(native, sc, cam, W, H, plan, kw, "brute", flags=native.BRUTE_MIXED_PACKETS, keep=(2,))
    assert v == VARIANT_BRUTE_WAVEFRONT and np.array_equal(_bits(plain[2]), _bits(early[2])) and np.array_equal(_bits(early[2]), _bits(mega[2]))


def test_brute_wavefront_full_hd_headline_configuration(native, oracle):
    """BASELINE config 1 (default scene, 1920x1080, 1 spp, 8 bounces) through the wavefront form, three frames as one launch:
    equal to the megakernel and to the CPU oracle's brute-force traversal, bit for bit."""
    from rvpt_amd import Camera
    W, H = 1920, 1080
    sc = scene_by_name("default")
    cam = Camera(W / H).get_data()
    kw = dict(max_bounces=8, aa=1)
    wave, st_w, v_w = _render(native, sc, cam, W, H, [(0, 3)], kw, "brute", flags=native.BRUTE_WAVEFRONT)
    mega, st_m, v_m = _render(native, sc, cam, W, H, [(0, 3)], kw, "brute", flags=native.BRUTE_MIXED_PACKETS)
    assert v_w == VARIANT_BRUTE_WAVEFRONT and v_m == 0
    assert np.array_equal(_bits(wave[2]), _bits(mega[2])) and tuple(st_w) == tuple(st_m)
    tris, mats, nodes = sc
    prev = None
    for f in range(3):
        prev, _ = oracle.render(oracle.settings_bytes(max_bounces=8, aa=1, current_frame=f), cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BRUTE, prev=prev)
    assert np.array_equal(_bits(wave[2]), _bits(prev))


def test_wavefront_flags_exclude_each_other(native):
    with pytest.raises(native.NativeError):
        native.Context(64, 64, 0, 0, 1, native.TRAVERSAL_BVH | native.BVH_WAVEFRONT | native.BVH_MEGAKERNEL)
