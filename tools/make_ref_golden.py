#!/usr/bin/env python3
"""Golden vectors from the reference's own compiled shader -> tests/golden/ref_spv/*.npz (authoring container only).

The generator is oracle/_ref/libref_spv*.so: /root/reference/assets/shaders/compute_pass.comp.spv — the binary the
reference loads (rvpt.cpp:676-681) — translated instruction by instruction to C by tools/spv2c.py and compiled by
oracle/ref_spv/Makefile.  Nothing of the reference travels with the fixtures: they hold inputs (scene arrays, camera
block, settings) and the images the shader produced for them.

Two images per case:
  `u`  the module executed with no floating-point contraction (libref_spv.so);
  `c`  executed under the build's contraction rule (spv2c.py --contract, libref_spv_fused.so) — the arithmetic the
       product implements, so the CPU oracle and the HIP path are compared with it BIT FOR BIT.
Images are RGB float32 (the shader stores alpha 0), frames 0 and 3 of an aa=2 accumulation chain through float images;
`q*` cases run the chain through rgba8 images as the reference does (uint8 texels).
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from oracle import oracle  # noqa: E402  (settings_bytes only)
from oracle.ref_spv import ref_spv  # noqa: E402
from rvpt_amd import native, scene  # noqa: E402
import make_golden as mg  # noqa: E402  (camera poses)

OUT = ROOT / "tests" / "golden" / "ref_spv"
SCENES = {"default": scene.default_scene, "showcase": scene.materials_showcase_scene, "cornell": lambda: scene.cornell_scene(1)}
MODES = list(range(11))  # 0-9 + 10 = eval_integrator's default branch (integrator_Hart)
FRAMES_KEPT = (0, 3)


def scene_arrays(name):
    tris, mats = SCENES[name]()
    nodes, idx = native.build_bvh(tris)
    return np.ascontiguousarray(tris[idx]), np.ascontiguousarray(mats), np.ascontiguousarray(nodes)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def chain(settings_kw, cam, sc, W, H, fused, frames=4, unorm8=False):
    tris, mats, nodes = sc
    prev, out = None, {}
    for f in range(frames):
        s = oracle.settings_bytes(current_frame=f, **settings_kw)
        prev = ref_spv.render(s, cam, nodes, tris, mats, W, H, prev=prev, unorm8=unorm8, fused=fused)
        out[f] = prev
    return out


def function_vectors():
    """Single functions of the module on inputs chosen to land ON decision boundaries (values such as the barycentrics or
    the slab test's booleans never reach a pixel of a small image).  -> functions.npz: inputs + outputs, both variants."""
    rng = np.random.RandomState(20260928)
    f32 = np.float32
    d = {}
    # --- intersect_triangle_fast: default-scene triangles + random ones; rays aimed at edge / vertex / interior points
    tris = np.concatenate([scene.default_scene()[0][::6], scene.make_triangles(rng.uniform(-2, 2, (40, 3, 3)), 0)]).astype(f32)
    cases = []
    for t in tris:
        v0, v1, v2 = t[0:3], t[4:7], t[8:11]
        for k in range(24):
            r = f32(rng.uniform(0.05, 0.95))
            u, v = [(0, r), (r, 0), (r, 1 - r), (0, 0), (1, 0), (0, 1), (r * 0.5, r * 0.4), (1.2 * r, 0.9)][k % 8]
            target = v0 + f32(u) * (v1 - v0) + f32(v) * (v2 - v0)
            org = (target + rng.uniform(-3, 3, 3)).astype(f32)
            if k % 5 == 0:
                org = f32([0, 0, 0]) if k % 10 == 0 else rng.randint(-2, 3, 3).astype(f32)  # exact, axis-friendly origins
            dirv = (target - org).astype(f32)
            if k % 3 == 0:
                dirv = (dirv / f32(max(np.linalg.norm(dirv), 1e-20))).astype(f32)
            if k == 23:
                dirv = (v1 - v0).astype(f32)  # in the plane: a zero (or cancelling) denominator
            cases.append(np.concatenate([org, dirv, v0, v1, v2]))
    cases = np.array(cases, f32)
    d["tri_in"] = cases
    for tag, fused in (("u", False), ("c", True)):
        out = np.zeros((len(cases), 10), f32)
        for i, c in enumerate(cases):
            acc, t, uv, nrm, pos = ref_spv.fn_intersect_triangle_fast(c[0:3], c[3:6], c[6:9], c[9:12], c[12:15], fused=fused)
            out[i] = [acc, t, uv[0], uv[1], *nrm, *pos]
        d[f"tri_{tag}"] = out
    # --- intersect_aabb: zero direction components, origins on the slab planes, inverted intervals
    boxes = []
    for k in range(600):
        lo = rng.uniform(-2, 1, 3)
        hi = lo + rng.uniform(0, 2, 3)
        org = rng.uniform(-3, 3, 3)
        dirv = rng.uniform(-1, 1, 3)
        if k % 10 < 7:  # aimed at (or just past a face / edge of) the box
            aim = lo + rng.choice([0.0, 1.0, 0.5, rng.uniform()], 3) * (hi - lo)
            dirv = aim - org
        if k % 4 == 0:
            dirv[rng.randint(3)] = 0.0
        if k % 8 == 0:
            dirv[rng.randint(3)] = -0.0
        if k % 6 == 0:
            a = rng.randint(3)
            org[a] = (lo if k % 12 == 0 else hi)[a]
        mint, maxt = (0.0, np.inf) if k % 3 else (0.0, rng.uniform(0.1, 4))
        boxes.append([*org, *dirv, *lo, *hi, mint, maxt])
    boxes = np.array(boxes, f32)
    d["aabb_in"] = boxes
    for tag, fused in (("u", False), ("c", True)):
        d[f"aabb_{tag}"] = np.array([ref_spv.fn_intersect_aabb(b[0:3], b[3:6], b[6:9], b[9:12], b[12], b[13], fused=fused) for b in boxes], np.uint8)
    # --- Fresnel, sphere mapping, sphere-traced distance
    fr = np.stack([rng.uniform(0, 1, 400), rng.uniform(0, 1, 400), rng.choice([1.5, 1 / 1.5, 1.33, 1 / 1.33, 2.4, 1.0], 400)], 1).astype(f32)
    d["fresnel_in"] = fr
    uv = np.concatenate([rng.uniform(0, 1, (400, 2)), [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [0.25, 1.0], [1.0, 0.5]]]).astype(f32)
    d["sphere_in"] = uv
    dt = rng.uniform(-2, 2, (300, 12)).astype(f32)
    d["dist_in"] = dt
    for tag, fused in (("u", False), ("c", True)):
        d[f"fresnel_{tag}"] = np.array([ref_spv.fn_fresnel(*x, fused=fused) for x in fr], f32)
        d[f"sphere_{tag}"] = np.array([ref_spv.fn_map_uniform_sphere(*x, fused=fused) for x in uv], f32)
        d[f"dist_{tag}"] = np.array([ref_spv.fn_distance_triangle(x[0:3], x[3:6], x[6:9], x[9:12], fused=fused) for x in dt], f32)
    # --- the three cameras (camera.glsl:29-99) over a grid of film coordinates, three poses
    cams, xy = [], np.array([(x, y) for x in np.linspace(0, 1, 7) for y in np.linspace(0, 1, 5)], f32)
    for pose in mg.CAMERAS:
        cams.append(mg.camera_block(pose))
    d["camera_blocks"], d["camera_xy"] = np.array(cams, f32), xy
    for tag, fused in (("u", False), ("c", True)):
        out = np.zeros((3, len(cams), len(xy), 6), f32)
        for ki, kind in enumerate(("pinhole", "ortho", "spherical")):
            for ci, cam in enumerate(cams):
                for j, (x, y) in enumerate(xy):
                    o, dd = ref_spv.fn_camera_ray(kind, cam, x, y, fused=fused)
                    out[ki, ci, j] = [*o, *dd]
        d[f"camera_{tag}"] = out
    # --- RNG: wang_hash and rand() streams (integer arithmetic + one conversion; identical in both variants)
    seeds = np.concatenate([[0, 1, 61, 0xFFFFFFFF, 0x80000000, 1920 * 1080 - 1], rng.randint(0, 2 ** 32, 58, dtype=np.uint64)]).astype(np.uint32)
    d["seeds"] = seeds
    d["wang"] = np.array([ref_spv.fn_wang_hash(s) for s in seeds], np.uint32)
    d["rand"] = np.array([ref_spv.fn_rand_stream(s if s else 1, 32) for s in seeds], f32)
    assert np.array_equal(d["rand"], np.array([ref_spv.fn_rand_stream(s if s else 1, 32, fused=True) for s in seeds], f32))
    np.savez_compressed(OUT / "functions.npz", **d)
    print("functions.npz:", len(cases), "triangle cases,", int(d["tri_u"][:, 0].sum()), "accepted;", len(boxes), "slab cases,", int(d["aabb_u"].sum()), "hit")


def big_cases():
    """The large configurations (BASELINE C3 / C4 / C5 shapes) executed by the reference binary.  The scenes are the build's own
    procedural ones (64 MB of triangles for the terrain): the fixture holds their sha256, the camera and the images — a test
    regenerates the scene and must arrive at the same bytes before it compares anything.
      big_terrain1m.npz   1 002 528-triangle heightfield (tree height > the 8 stack levels the HIP kernel keeps in LDS: overflow
                          levels, packed stack heads), 64x48, aa 2, frames 0 and 3 of a 4-frame chain
      big_cornell.npz     Cornell box + 9 152-triangle model (C3's scene), 160x96, aa 2, frames 0 and 3
      big_cornell_16spp.npz  the same scene, 64x48, aa 16, an 8-frame chain (C5's shape: compute_pass.comp:146-166), frames 0, 3, 7
    (heights are multiples of 16: the reference dispatches H/16 groups with integer division and leaves the other rows unwritten,
    rvpt.cpp:1035-1036)"""
    from rvpt_amd import Camera

    def cam_block(aspect, translation, rotation=(0.0, 0.0, 0.0)):
        c = Camera(aspect)
        c.translation = np.array(translation, dtype=np.float64)
        c.rotation = np.array(rotation, dtype=np.float64)
        return c.get_data()

    def arrays(tris, mats):
        nodes, idx = native.build_bvh(tris)
        return np.ascontiguousarray(tris[idx]), np.ascontiguousarray(mats), np.ascontiguousarray(nodes)

    jobs = [("big_terrain1m", scene.heightfield_scene, cam_block(64 / 48, (0.0, 2.5, -5.0), (0.0, 25.0, 0.0)), 64, 48, 2, 4, (0, 3)),
            ("big_cornell", scene.cornell_scene, cam_block(160 / 96, (0.0, 2.0, -1.9)), 160, 96, 2, 4, (0, 3)),
            ("big_cornell_16spp", scene.cornell_scene, cam_block(64 / 48, (0.0, 2.0, -1.9)), 64, 48, 16, 8, (0, 3, 7))]
    for name, make, cam, W, H, aa, n_frames, keep in jobs:
        sc = arrays(*make())
        data = {"camera": cam, "aa": aa, "max_bounces": 8, "width": W, "height": H, "frames": n_frames, "n_tris": sc[0].shape[0],
                "bvh_nodes": sc[2].nbytes // 32, "scene_sha256": digest(*sc)}
        for tag, fused in (("u", False), ("c", True)):
            imgs = chain(dict(max_bounces=8, aa=aa), cam, sc, W, H, fused, frames=n_frames)
            for f in keep:
                assert not imgs[f][..., 3].any()
                data[f"f{f}_{tag}"] = imgs[f][..., :3].copy()
        np.savez_compressed(OUT / f"{name}.npz", **data)
        print(name, sc[0].shape[0], "triangles,", data["bvh_nodes"], "nodes written")


def converged_case():
    """A DIFFERENT admissible execution of the reference binary (libref_spv_libm.so: no contraction, IEEE quotient, libm
    sin/cos/tan, dot products summed the other way round, normalize by division) run to convergence: 256 accumulation frames
    of the default scene at 64x32, 1 spp.  Its single frames differ from the product's pixel by pixel (a path tracer is
    chaotic); its MEAN must agree with the product's within Monte-Carlo error.  -> converged_libm.npz: the mean and the
    per-pixel variance of the 256 per-frame sample images (float64 statistics of float32 images)."""
    sc = scene_arrays("default")
    tris, mats, nodes = sc
    cam = mg.camera_block("bench")
    W, H, N = 64, 32, 256
    total = np.zeros((H, W, 3), np.float64)
    total_sq = np.zeros((H, W, 3), np.float64)
    for f in range(N):
        # frame f with an empty temporal image: out = (0*f + sampled) / (f + 1)  ->  sampled = out * (f + 1) (exact enough for statistics)
        s = oracle.settings_bytes(max_bounces=8, aa=1, current_frame=f)
        img = ref_spv.render(s, cam, nodes, tris, mats, W, H, prev=None, fused="libm")[..., :3].astype(np.float64) * (f + 1)
        total += img
        total_sq += img * img
    mean = total / N
    var = np.maximum(total_sq / N - mean * mean, 0.0) * N / (N - 1)
    np.savez_compressed(OUT / "converged_libm.npz", camera=cam, aa=1, max_bounces=8, width=W, height=H, frames=N,
                        mean=mean.astype(np.float32), var=var.astype(np.float32))
    print("converged_libm.npz: mean", mean.mean(), "mean var", var.mean())


def main():
    ref_spv.build()
    OUT.mkdir(parents=True, exist_ok=True)
    if "--only" in sys.argv:  # add / refresh one group of fixtures without touching the others
        what = sys.argv[sys.argv.index("--only") + 1]
        {"big": big_cases, "converged": converged_case, "functions": function_vectors}[what]()
        return
    for old in OUT.glob("*.npz"):
        old.unlink()
    L = ref_spv.lib()
    print("module:", L.ref_spv_function_count(), "functions,", L.ref_spv_instruction_count(), "instructions in blocks")
    for sname in SCENES:
        sc = scene_arrays(sname)
        tris, mats, nodes = sc
        np.savez_compressed(OUT / f"scene_{sname}.npz", tris=tris, mats=mats, nodes=nodes.view(np.uint32).reshape(-1, 8),
                            sha256=digest(tris, mats, nodes))
        poses = list(mg.CAMERAS) if sname != "cornell" else ["bench"]
        for pose in poses:
            cam = mg.camera_block(pose)
            for cmode in ((0, 1, 2) if pose == "oblique" else (0,)):
                data = {"camera": cam, "camera_mode": cmode, "aa": 2, "max_bounces": 8, "scene_sha256": digest(*sc)}
                for mode in (MODES if sname != "cornell" else [9]):
                    W, H = (64, 32) if mode == 9 else (32, 16)
                    kw = dict(max_bounces=8, aa=2, camera_mode=cmode, modes=(mode,) * 4)
                    for tag, fused in (("u", False), ("c", True)):
                        imgs = chain(kw, cam, sc, W, H, fused)
                        for f in FRAMES_KEPT:
                            assert not imgs[f][..., 3].any()
                            data[f"m{mode}_f{f}_{tag}"] = imgs[f][..., :3].copy()
                np.savez_compressed(OUT / f"{sname}_{pose}_cam{cmode}.npz", **data)
                print(sname, pose, "camera_mode", cmode, "written")
    # one larger Kajiya image (more rays on triangle edges and box faces than 64x32 offers) and a deeper tree (24x24-cell terrain)
    sc = scene_arrays("default")
    cam = mg.camera_block("bench")
    data = {"camera": cam, "aa": 1, "max_bounces": 8}
    for tag, fused in (("u", False), ("c", True)):
        data[f"f0_{tag}"] = chain(dict(max_bounces=8, aa=1), cam, sc, 256, 128, fused, frames=1)[0][..., :3].copy()
    np.savez_compressed(OUT / "large_default_bench.npz", **data)
    tris, mats = scene.heightfield_scene(cells=24)
    nodes, idx = native.build_bvh(tris)
    sc = (np.ascontiguousarray(tris[idx]), np.ascontiguousarray(mats), np.ascontiguousarray(nodes))
    np.savez_compressed(OUT / "scene_terrain24.npz", tris=sc[0], mats=sc[1], nodes=sc[2].view(np.uint32).reshape(-1, 8), sha256=digest(*sc))
    from rvpt_amd import Camera
    c = Camera(2.0)
    c.translation = np.array([0.0, 2.5, -5.0])
    c.rotation = np.array([0.0, 25.0, 0.0])
    cam_t = c.get_data()
    data = {"camera": cam_t, "aa": 2, "max_bounces": 8}
    for tag, fused in (("u", False), ("c", True)):
        imgs = chain(dict(max_bounces=8, aa=2), cam_t, sc, 64, 32, fused)
        for f in FRAMES_KEPT:
            data[f"f{f}_{tag}"] = imgs[f][..., :3].copy()
    np.savez_compressed(OUT / "terrain24_kajiya.npz", **data)
    # split screen (compute_pass.comp:134-144): four different integrators, off-centre split
    sc = scene_arrays("showcase")
    cam = mg.camera_block("bench")
    data = {"camera": cam, "modes": np.array([9, 5, 7, 10]), "split": np.array([0.4, 0.6], np.float32), "aa": 2, "max_bounces": 8}
    for tag, fused in (("u", False), ("c", True)):
        imgs = chain(dict(max_bounces=8, aa=2, modes=(9, 5, 7, 10), split=(0.4, 0.6)), cam, sc, 64, 32, fused)
        for f in FRAMES_KEPT:
            data[f"f{f}_{tag}"] = imgs[f][..., :3].copy()
    np.savez_compressed(OUT / "split_showcase_bench.npz", **data)
    # bounce budget exhausted -> black (integrators.glsl:674-675), one sample per pixel
    data = {"camera": cam, "aa": 1, "max_bounces": 2}
    for tag, fused in (("u", False), ("c", True)):
        imgs = chain(dict(max_bounces=2, aa=1), cam, sc, 64, 32, fused)
        for f in FRAMES_KEPT:
            data[f"f{f}_{tag}"] = imgs[f][..., :3].copy()
    np.savez_compressed(OUT / "bounces2_showcase_bench.npz", **data)
    # the reference's real storage: rgba8 temporal + output images (compute_pass.comp:41-42), 6-frame chain
    sc = scene_arrays("default")
    data = {"camera": cam, "aa": 1, "max_bounces": 8}
    for tag, fused in (("u", False), ("c", True)):
        imgs = chain(dict(max_bounces=8, aa=1), cam, sc, 64, 32, fused, frames=6, unorm8=True)
        for f in (0, 1, 5):
            q = np.rint(imgs[f] * 255.0).astype(np.uint8)
            assert np.array_equal(q.astype(np.float32) / np.float32(255.0), imgs[f])
            data[f"q{f}_{tag}"] = q
    np.savez_compressed(OUT / "unorm8_default_bench.npz", **data)
    function_vectors()
    big_cases()
    converged_case()
    total = sum(p.stat().st_size for p in OUT.glob("*.npz"))
    print(f"{len(list(OUT.glob('*.npz')))} files, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
