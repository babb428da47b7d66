"""Shared reader of the tests/golden/ref_spv fixtures (made by tools/make_ref_golden.py from the reference's compiled
shader).  Yields (case id, scene arrays, camera, settings keywords, width, height, {frame: {tag: rgb image}})."""
import hashlib

import numpy as np

from _util import GOLDEN

REF = GOLDEN / "ref_spv"


def load_scene(name):
    z = np.load(REF / f"scene_{name}.npz")
    tris, mats, nodes = z["tris"], z["mats"], np.ascontiguousarray(z["nodes"]).view(np.uint8).reshape(-1)
    h = hashlib.sha256()
    for a in (tris, mats, nodes):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(z["sha256"])
    return tris, mats, nodes


def mode_cases():
    """Every (scene, pose, camera_mode, integrator mode) case: one entry per mode of every per-camera file."""
    out = []
    for p in sorted(REF.glob("*_cam?.npz")):
        z = np.load(p)
        modes = sorted({int(k.split("_")[0][1:]) for k in z.files if k.startswith("m") and k[1].isdigit()})
        for m in modes:
            out.append((p.stem, m))
    return out


def load_mode_case(stem, mode):
    z = np.load(REF / f"{stem}.npz")
    sname = stem.split("_")[0]
    frames = {}
    for f in (0, 3):
        frames[f] = {t: z[f"m{mode}_f{f}_{t}"] for t in ("u", "c")}
    H, W = frames[0]["u"].shape[:2]
    kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]), camera_mode=int(z["camera_mode"]), modes=(mode,) * 4)
    return load_scene(sname), z["camera"], kw, W, H, frames
