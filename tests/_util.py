"""Shared helpers for the test modules."""
import math
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


def rel_l2(a, b) -> float:
    """Relative L2 distance of two images (north_star bar: <= 1e-4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = math.sqrt(float((b * b).sum()))
    return math.sqrt(float(((a - b) ** 2).sum())) / (den if den > 0 else 1.0)


def identity_camera(aspect, fov_deg=90.0):
    cam = np.zeros(20, np.float32)
    cam[[0, 5, 10, 15]] = 1.0
    cam[16], cam[17], cam[18] = aspect, math.radians(fov_deg), 4.0
    return cam


def scene_by_name(name):
    """(sorted_tris, mats, nodes) in BVH-leaf order for a named synthetic scene."""
    from rvpt_amd import native, scene
    make = {"default": scene.default_scene, "showcase": scene.materials_showcase_scene,
            "cornell": scene.cornell_scene}[name]
    tris, mats = make()
    nodes, idx = native.build_bvh(tris)
    return tris[idx], mats, nodes
