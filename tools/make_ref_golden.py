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


def main():
    ref_spv.build()
    OUT.mkdir(parents=True, exist_ok=True)
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
    total = sum(p.stat().st_size for p in OUT.glob("*.npz"))
    print(f"{len(list(OUT.glob('*.npz')))} files, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
