#!/usr/bin/env python3
"""Generate tests/golden/*.npz with the CPU oracle (run in the authoring container).

The reference ships no golden images, so these fixtures pin the ORACLE at the commit that
created them: any later change to oracle/rvpt_oracle.c that alters a bit fails tests/test_oracle_golden.py.
Inputs are regenerated from rvpt_amd.scene at test time; only camera/settings + expected output live here.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402
from rvpt_amd import Camera, native, scene  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)
W, H = 64, 32

CAMERAS = {
    "default": dict(translation=(0, 0, 0), rotation=(0, 0, 0)),       # camera.h:44-53 defaults
    "bench": dict(translation=(0, 0.9, -2.5), rotation=(0, 0, 0)),    # frames the whole model
    "oblique": dict(translation=(1.5, 1.2, -1.8), rotation=(-35, 10, 0)),
}
SCENES = {"default": scene.default_scene, "showcase": scene.materials_showcase_scene}


def camera_block(name):
    c = Camera(W / H)
    c.translation = np.array(CAMERAS[name]["translation"], dtype=np.float64)
    c.rotation = np.array(CAMERAS[name]["rotation"], dtype=np.float64)
    return c.get_data()


def main():
    for sname, make in SCENES.items():
        tris, mats = make()
        nodes, idx = native.build_bvh(tris)
        st = tris[idx]
        for cname in CAMERAS:
            cam = camera_block(cname)
            travs = [("brute", oracle.TRAVERSAL_BRUTE), ("bvh", oracle.TRAVERSAL_BVH)]
            if cname == "oblique":  # the opt-in ordered traversal (build-defined, oracle traversal 2): one camera per scene
                travs.append(("bvhordered", oracle.TRAVERSAL_BVH_ORDERED))
            for trav_name, trav in travs:
                prev = None
                frames = {}
                for f in range(4):  # frames 0..3, aa=2, temporal chain
                    s = oracle.settings_bytes(max_bounces=8, aa=2, current_frame=f)
                    img, stats = oracle.render(s, cam, nodes, st, mats, W, H, trav, prev=prev)
                    prev = img
                    if f in (0, 3):
                        frames[f"frame{f}"] = img
                        frames[f"stats{f}"] = stats
                np.savez_compressed(OUT / f"{sname}_{cname}_{trav_name}.npz", camera=cam, width=W, height=H, aa=2,
                                    max_bounces=8, **frames)
                print(sname, cname, trav_name, frames["stats0"], float(frames["frame3"].mean()))


if __name__ == "__main__":
    main()
