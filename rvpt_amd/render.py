"""Headless renderer:  python -m rvpt_amd.render --scene default --width 1920 --height 1080 --frames 64 --out a.png

The reference has no CLI (window + ImGui sliders, rvpt.cpp:276-277); the flags map one-to-one onto
RenderSettings (rvpt.h:77-89) and the Camera fields (camera.h:44-53)."""
from __future__ import annotations

import argparse
import json
import time

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--scene", default="default", help="default | cornell | showcase | heightfield | path to a .obj")
    ap.add_argument("--materials-from-mtl", action="store_true",
                    help="with a .obj path: take the materials from its mtllib/usemtl records instead of the demo's constant material id")
    ap.add_argument("--width", type=int, default=1024)   # Window::Settings in main.cpp:95-98
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=1, help="RenderSettings.aa")
    ap.add_argument("--bounces", type=int, default=8, help="RenderSettings.max_bounces")
    ap.add_argument("--frames", type=int, default=16, help="temporally accumulated frames")
    ap.add_argument("--traversal", choices=["brute", "bvh", "bvh_ordered"], default="bvh")
    ap.add_argument("--translate", type=float, nargs=3, default=(0.0, 0.0, 0.0))
    ap.add_argument("--rotate", type=float, nargs=3, default=(0.0, 0.0, 0.0), help="degrees, as Camera::rotate")
    ap.add_argument("--fov", type=float, default=90.0)
    ap.add_argument("--unorm8-accum", action="store_true", help="accumulate through 8 bits like the reference")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="frame.png", help=".png (UNORM8) or .pfm (float)")
    a = ap.parse_args(argv)

    from . import RVPT, imageio, native, scene
    if a.scene == "default":
        tris, mats = scene.default_scene()
    elif a.scene == "cornell":
        tris, mats = scene.cornell_scene()
    elif a.scene == "showcase":
        tris, mats = scene.materials_showcase_scene()
    elif a.scene == "heightfield":
        tris, mats = scene.heightfield_scene()
    elif a.materials_from_mtl:  # OBJ + MTL scene description (usemtl / mtllib)
        tris, mats, _ = scene.load_obj_scene(a.scene)
    else:  # load_model(path, 1) + the two demo materials, main.cpp:102-107
        tris, mats = scene.make_triangles(scene.load_obj_positions(a.scene), 1), scene.default_materials()

    r = RVPT(a.width, a.height, device=a.device, traversal=a.traversal,
             flags=native.TIMING | (native.ACCUM_UNORM8 if a.unorm8_accum else 0))
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.render_settings.aa, r.render_settings.max_bounces = a.spp, a.bounces
    r.scene_camera.translation = np.asarray(a.translate, dtype=np.float64)
    r.scene_camera.rotation = np.asarray(a.rotate, dtype=np.float64)
    r.scene_camera.set_fov(a.fov)
    t0 = time.perf_counter()
    r.initialize()
    t1 = time.perf_counter()
    for _ in range(a.frames):
        r.update()
        r.draw()
    r.wait()
    t2 = time.perf_counter()
    if a.out.endswith(".pfm"):
        imageio.write_pfm(a.out, r.read_frame(native.FORMAT_RGBA32F))
    else:
        imageio.write_png(a.out, r.read_frame(native.FORMAT_RGBA8_UNORM))
    print(json.dumps({"out": a.out, "triangles": int(tris.shape[0]), "init_s": round(t1 - t0, 3),
                      "render_s": round(t2 - t1, 4), "Msamples_per_s": round(a.width * a.height * a.spp * a.frames / (t2 - t1) / 1e6, 1)}))
    r.shutdown()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
