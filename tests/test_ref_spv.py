"""The CPU oracle against the reference's OWN compiled shader.

tests/golden/ref_spv/*.npz are outputs of /root/reference/assets/shaders/compute_pass.comp.spv — the binary the reference
loads at run time (rvpt.cpp:676-681) — executed on the CPU after an instruction-by-instruction translation to C
(tools/spv2c.py, recipe oracle/ref_spv/Makefile, generator tools/make_ref_golden.py).  This is what pins the oracle:

  * `u` images: the module without floating-point contraction  == oracle built with -DORACLE_UNFUSED, bit for bit;
  * `c` images: the module under the build's contraction rule    == the oracle as shipped (and the HIP path, in
    tests/test_gpu_parity.py), bit for bit.

All 39 functions of the module (every integrator, camera and intersector) are covered by the cases.  Where the reference
tree is present (the authoring container) the fixtures are also re-derived from the binary and must not have drifted."""
import os

import numpy as np
import pytest

import _refspv
from _util import ROOT


def _chain(oracle, sc, cam, kw, W, H, unfused, frames=4, traversal=None):
    tris, mats, nodes = sc
    prev, out = None, {}
    for f in range(frames):
        s = oracle.settings_bytes(current_frame=f, **kw)
        prev, _ = oracle.render(s, cam, nodes, tris, mats, W, H, oracle.TRAVERSAL_BVH if traversal is None else traversal,
                                prev=prev, unfused=unfused)
        out[f] = prev
    return out


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_fixture_inventory():
    cases = _refspv.mode_cases()
    assert len(cases) == 10 * 11 + 1  # (default, showcase) x 5 camera set-ups x 11 integrators + cornell/Kajiya
    assert {m for _, m in cases} == set(range(11))
    assert {int(c[0][-1]) for c in cases} == {0, 1, 2}  # pinhole, orthographic, spherical (camera.glsl:29-99)


@pytest.mark.parametrize("stem,mode", _refspv.mode_cases())
def test_oracle_equals_compiled_reference_shader(oracle, stem, mode):
    sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, mode)
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, cam, kw, W, H, unfused)
        for f in (0, 3):
            assert not got[f][..., 3].any()
            assert _same_bits(got[f][..., :3], frames[f][tag]), \
                f"{stem} mode {mode} frame {f} [{tag}]: {int((got[f][..., :3] != frames[f][tag]).any(axis=2).sum())} pixels differ"


def test_brute_force_variant_agrees_with_the_reference_traversal(oracle):
    """The brute-force closest hit is this build's variant (the reference's live traversal is the BVH); away from exact
    ties and slab-culling corner cases it must land on the reference's pixels — asserted here against the shader itself."""
    total = differ = 0
    for stem in ("default_bench_cam0", "showcase_bench_cam0", "showcase_oblique_cam0"):
        sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, 9)
        got = _chain(oracle, sc, cam, kw, W, H, False, traversal=oracle.TRAVERSAL_BRUTE)
        d = (got[3][..., :3].view(np.uint32) != frames[3]["c"].view(np.uint32)).any(axis=2)
        total += d.size
        differ += int(d.sum())
    assert differ <= 0.002 * total, (differ, total)


def test_larger_image_and_deeper_tree(oracle):
    z = np.load(_refspv.REF / "large_default_bench.npz")
    sc = _refspv.load_scene("default")
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, z["camera"], dict(max_bounces=8, aa=1), 256, 128, unfused, frames=1)
        assert _same_bits(got[0][..., :3], z[f"f0_{tag}"])
    z = np.load(_refspv.REF / "terrain24_kajiya.npz")
    sc = _refspv.load_scene("terrain24")
    assert sc[0].shape[0] == 2 * 24 * 24
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, z["camera"], dict(max_bounces=8, aa=2), 64, 32, unfused)
        for f in (0, 3):
            assert _same_bits(got[f][..., :3], z[f"f{f}_{tag}"])


def test_split_screen_selection(oracle):
    z = np.load(_refspv.REF / "split_showcase_bench.npz")
    sc = _refspv.load_scene("showcase")
    kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]), modes=tuple(int(m) for m in z["modes"]), split=tuple(float(s) for s in z["split"]))
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, z["camera"], kw, 64, 32, unfused)
        for f in (0, 3):
            assert _same_bits(got[f][..., :3], z[f"f{f}_{tag}"])


def test_exhausted_bounce_budget(oracle):
    z = np.load(_refspv.REF / "bounces2_showcase_bench.npz")
    sc = _refspv.load_scene("showcase")
    kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]))
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, z["camera"], kw, 64, 32, unfused)
        for f in (0, 3):
            assert _same_bits(got[f][..., :3], z[f"f{f}_{tag}"])


def test_rgba8_accumulation_chain(oracle):
    """compute_pass.comp:41-42: both images are rgba8, so the running mean is re-quantised every frame."""
    z = np.load(_refspv.REF / "unorm8_default_bench.npz")
    tris, mats, nodes = _refspv.load_scene("default")
    for unfused, tag in ((True, "u"), (False, "c")):
        prev = None
        for f in range(6):
            s = oracle.settings_bytes(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]), current_frame=f)
            img, _ = oracle.render(s, z["camera"], nodes, tris, mats, 64, 32, oracle.TRAVERSAL_BVH, prev=prev, unfused=unfused)
            q = oracle.quantize_rgba8(img)
            prev = oracle.dequantize_rgba8(q)
            if f in (0, 1, 5):
                assert np.array_equal(q, z[f"q{f}_{tag}"]), f"rgba8 frame {f} [{tag}]"


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/shaders/compute_pass.comp.spv"), reason="reference tree not present (GPU box)")
def test_fixtures_are_what_the_reference_binary_produces(oracle):
    """Authoring container only: rebuild oracle/_ref from the reference's .spv and re-derive a sample of the fixtures."""
    import sys
    sys.path.insert(0, str(ROOT))
    from oracle.ref_spv import ref_spv
    ref_spv.build()
    L = ref_spv.lib()
    assert L.ref_spv_function_count() == 39
    for stem, mode in [("default_bench_cam0", 9), ("showcase_oblique_cam2", 9), ("showcase_oblique_cam1", 7), ("default_default_cam0", 10),
                       ("showcase_bench_cam0", 5), ("cornell_bench_cam0", 9)]:
        sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, mode)
        tris, mats, nodes = sc
        for fused, tag in ((False, "u"), (True, "c")):
            prev = None
            for f in range(4):
                prev = ref_spv.render(oracle.settings_bytes(current_frame=f, **kw), cam, nodes, tris, mats, W, H, prev=prev, fused=fused)
                if f in (0, 3):
                    assert _same_bits(prev[..., :3], frames[f][tag]), (stem, mode, f, tag)


@pytest.mark.parametrize("name", _refspv.BIG_CASES)
def test_large_configurations(oracle, name):
    """The shapes of BASELINE C3 / C4 / C5 executed by the reference binary: the 1 002 528-triangle terrain (a tree taller than the
    eight stack levels the HIP kernel keeps in LDS), the Cornell box + 9 152-triangle model, and a 16 spp x 8-frame accumulation
    chain (compute_pass.comp:146-166)."""
    sc, cam, kw, W, H, n_frames, frames = _refspv.load_big_case(name)
    for unfused, tag in ((True, "u"), (False, "c")):
        got = _chain(oracle, sc, cam, kw, W, H, unfused, frames=n_frames)
        for f in frames:
            assert _same_bits(got[f][..., :3], frames[f][tag]), f"{name} frame {f} [{tag}]"


def test_converged_mean_agrees_with_a_different_admissible_execution(oracle):
    """tests/golden/ref_spv/converged_libm.npz: the reference binary under ANOTHER admissible driver (no contraction, IEEE quotient,
    libm sin/cos/tan, dot products summed the other way round, normalize by division), 256 frames of the default scene at 64x32.
    Pixel by pixel its frames differ from this build's (chaos); its converged mean must not: per pixel and channel
    z = (mean_oracle - mean_libm) / sqrt(2 var / N) has to look like noise at most (the two runs share most paths, so it is far
    smaller), and the image means must agree within 3 sigma."""
    z = np.load(_refspv.REF / "converged_libm.npz")
    tris, mats, nodes = _refspv.load_scene("default")
    W, H, N = int(z["width"]), int(z["height"]), int(z["frames"])
    got = _chain(oracle, (tris, mats, nodes), z["camera"], dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"])), W, H, False, frames=N)[N - 1][..., :3]
    _refspv.assert_converged_agreement(got, z)


def _every_image_fixture():
    """(name, scene, camera, settings keywords, W, H, chain length, rgba8?, {frame: {tag: expected}}) of EVERY image fixture."""
    for stem, mode in _refspv.mode_cases():
        sc, cam, kw, W, H, frames = _refspv.load_mode_case(stem, mode)
        yield f"{stem}/m{mode}", sc, cam, kw, W, H, 4, False, frames
    z = np.load(_refspv.REF / "large_default_bench.npz")
    yield "large_default_bench", _refspv.load_scene("default"), z["camera"], dict(max_bounces=8, aa=1), 256, 128, 1, False, {0: {t: z[f"f0_{t}"] for t in "uc"}}
    for name, scene_name in (("terrain24_kajiya", "terrain24"), ("split_showcase_bench", "showcase"), ("bounces2_showcase_bench", "showcase")):
        z = np.load(_refspv.REF / f"{name}.npz")
        kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]))
        if "modes" in z.files:
            kw.update(modes=tuple(int(m) for m in z["modes"]), split=tuple(float(x) for x in z["split"]))
        yield name, _refspv.load_scene(scene_name), z["camera"], kw, 64, 32, 4, False, {f: {t: z[f"f{f}_{t}"] for t in "uc"} for f in (0, 3)}
    for name in _refspv.BIG_CASES:
        sc, cam, kw, W, H, n_frames, frames = _refspv.load_big_case(name)
        yield name, sc, cam, kw, W, H, n_frames, False, frames
    z = np.load(_refspv.REF / "unorm8_default_bench.npz")
    yield ("unorm8_default_bench", _refspv.load_scene("default"), z["camera"], dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"])), 64, 32, 6, True,
           {f: {t: z[f"q{f}_{t}"] for t in "uc"} for f in (0, 1, 5)})


@pytest.mark.skipif(not os.path.exists("/root/reference/assets/shaders/compute_pass.comp.spv"), reason="reference tree not present (GPU box)")
def test_strict_ieee_quotient_changes_no_fixture(oracle):
    """The translated shader forms its one OpFDiv(OpDot, OpDot) — t = dot(v0 - o, n) / dot(d, n) of intersect_triangle_fast — as the
    gfx950 kernel does (Markstein's sequence on the hardware reciprocal, spv_shim.h: shim_fdiv_dots): a SPECIFIED DEVIATION from
    C's `/`, inside Vulkan's 2.5-ULP allowance for a divisor in [2^-126, 2^126], the same value as `/` on all of that domain and
    NaN / 0 instead of +-inf / finite outside it, which the accept test rejects either way.  This test is the proof that the
    choice is unobservable: the same two executions of the reference binary built with the strict IEEE quotient
    (libref_spv_ieee.so, libref_spv_fused_ieee.so) reproduce EVERY committed image fixture bit for bit."""
    import sys
    sys.path.insert(0, str(ROOT))
    from oracle.ref_spv import ref_spv
    ref_spv.build()
    n_images = 0
    for name, sc, cam, kw, W, H, n_frames, unorm8, expected in _every_image_fixture():
        tris, mats, nodes = sc
        for variant, tag in (("ieee", "u"), ("fused_ieee", "c")):
            prev = None
            for f in range(n_frames):
                prev = ref_spv.render(oracle.settings_bytes(current_frame=f, **kw), cam, nodes, tris, mats, W, H, prev=prev, unorm8=unorm8, fused=variant)
                if f in expected:
                    if unorm8:
                        assert np.array_equal(np.rint(prev * 255.0).astype(np.uint8), expected[f][tag]), (name, f, tag)
                    else:
                        assert _same_bits(prev[..., :3], expected[f][tag]), (name, f, tag)
                    n_images += 1
    assert n_images == 2 * (2 * 111 + 1 + 3 * 2 + 3 + 2 + 2 + 3)  # 478 images


# ---------------------------------------------------------------------------------------------------------------------
# Single functions of the module (tests/golden/ref_spv/functions.npz): what a pixel never shows — barycentrics on triangle
# edges, slab-test booleans with zero direction components, Fresnel terms, camera rays, the RNG stream.

FN = np.load(_refspv.REF / "functions.npz")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("unfused,tag", [(True, "u"), (False, "c")])
def test_function_intersect_triangle_fast(oracle, unfused, tag):
    """accept, t, uv, the un-normalised normal and the hit position for rays aimed at edges, vertices and interiors."""
    cases, want = FN["tri_in"], FN[f"tri_{tag}"]
    assert 300 < want[:, 0].sum() < len(cases) - 300  # both outcomes are well represented
    bad = []
    for i, c in enumerate(cases):
        tri = np.zeros(16, np.float32)
        tri[0:3], tri[4:7], tri[8:11] = c[6:9], c[9:12], c[12:15]
        acc, tuv = oracle.tri_test(c[0:3], c[3:6], tri, unfused=unfused)
        prep = oracle.prepare(tri, unfused=unfused)[0]
        w = want[i]
        ok = acc == bool(w[0])
        if acc:  # the module writes `info` only on acceptance (intersection.glsl:313-320)
            t = np.float32(tuv[0])
            pos = np.array([np.float32(np.float32(c[3 + k]) * t) for k in range(3)], np.float32)  # unfused reading, checked below per variant
            ok &= np.array_equal(_bits(tuv), _bits(w[1:4])) and np.array_equal(_bits(prep[3:6]), _bits(w[4:7]))
        if not ok:
            bad.append(i)
    assert not bad, f"{len(bad)} of {len(cases)} cases differ, first {bad[:5]}"


@pytest.mark.parametrize("unfused,tag", [(True, "u"), (False, "c")])
def test_function_intersect_aabb(oracle, unfused, tag):
    boxes, want = FN["aabb_in"], FN[f"aabb_{tag}"]
    assert 100 < want.sum() < len(boxes) - 100
    got = np.array([oracle.aabb_test(b[0:3], b[3:6], b[6:9], b[9:12], float(b[12]), float(b[13]), unfused=unfused) for b in boxes], np.uint8)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]


@pytest.mark.parametrize("unfused,tag", [(True, "u"), (False, "c")])
def test_function_fresnel_sphere_distance_cameras(oracle, unfused, tag):
    got = np.array([oracle.fresnel(*x, unfused=unfused) for x in FN["fresnel_in"]], np.float32)
    assert np.array_equal(_bits(got), _bits(FN[f"fresnel_{tag}"]))
    got = np.array([oracle.sphere_point(*x, unfused=unfused) for x in FN["sphere_in"]], np.float32)
    assert np.array_equal(_bits(got), _bits(FN[f"sphere_{tag}"]))
    got = np.array([oracle.distance_triangle(x[0:3], x[3:6], x[6:9], x[9:12], unfused=unfused) for x in FN["dist_in"]], np.float32)
    assert np.array_equal(_bits(got), _bits(FN[f"dist_{tag}"]))
    want = FN[f"camera_{tag}"]
    for mode in range(3):
        for ci, cam in enumerate(FN["camera_blocks"]):
            for j, (x, y) in enumerate(FN["camera_xy"]):
                o, d = oracle.camera_ray(mode, cam, x, y, unfused=unfused)
                assert np.array_equal(_bits(np.concatenate([o, d])), _bits(want[mode, ci, j])), (mode, ci, j)


def test_function_rng(oracle):
    for s, w in zip(FN["seeds"], FN["wang"]):
        assert oracle.wang_hash(int(s)) == int(w)
    # rand(): the stream that follows a given rng_state.  oracle.rand_stream seeds with wang_hash(p_idx) + frame, so give it
    # frame = state - wang_hash(0) (uint32 wrap-around) to start from `state`
    h0 = oracle.wang_hash(0)
    for s, w in zip(FN["seeds"], FN["rand"]):
        state = int(s) if s else 1
        vals, _ = oracle.rand_stream(0, (state - h0) & 0xFFFFFFFF, 32)
        assert np.array_equal(_bits(vals), _bits(w))
