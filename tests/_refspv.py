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


BIG_CASES = ("big_terrain1m", "big_cornell", "big_cornell_16spp")
_BIG_SCENES = {}


def load_big_case(name):
    """A large configuration executed by the reference binary (tools/make_ref_golden.py::big_cases).  The scene is the build's own
    procedural one — regenerated here, and its bytes must hash to what the fixture was rendered from."""
    from rvpt_amd import native, scene
    z = np.load(REF / f"{name}.npz")
    kind = "terrain" if "terrain" in name else "cornell"
    if kind not in _BIG_SCENES:
        tris, mats = (scene.heightfield_scene if kind == "terrain" else scene.cornell_scene)()
        nodes, idx = native.build_bvh(tris)
        _BIG_SCENES[kind] = (np.ascontiguousarray(tris[idx]), np.ascontiguousarray(mats), np.ascontiguousarray(nodes))
    sc = _BIG_SCENES[kind]
    h = hashlib.sha256()
    for a in sc:
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == str(z["scene_sha256"]), f"{name}: the regenerated scene is not the one the fixture was rendered from"
    assert sc[0].shape[0] == int(z["n_tris"])
    keep = sorted(int(k[1:-2]) for k in z.files if k.endswith("_c"))
    frames = {f: {t: z[f"f{f}_{t}"] for t in ("u", "c")} for f in keep}
    kw = dict(max_bounces=int(z["max_bounces"]), aa=int(z["aa"]))
    return sc, z["camera"], kw, int(z["width"]), int(z["height"]), int(z["frames"]), frames


def assert_converged_agreement(mean_product, z):
    """mean_product [H,W,3]: this build's mean of the N frames; z: converged_libm.npz (mean, var of a different admissible execution)."""
    N = int(z["frames"])
    ref, var = z["mean"].astype(np.float64), z["var"].astype(np.float64)
    sigma = np.sqrt(2.0 * var / N)
    noisy = sigma > 0
    zscore = (mean_product.astype(np.float64) - ref)[noisy] / sigma[noisy]
    assert np.abs(mean_product[~noisy] - ref[~noisy]).max(initial=0.0) <= 1e-5  # zero-variance pixels (sky): the same value
    assert float(np.sqrt((zscore ** 2).mean())) <= 1.5, "per-pixel differences exceed Monte-Carlo noise"
    assert float(np.abs(zscore).max()) <= 6.0
    image_sigma = float(np.sqrt((sigma ** 2).sum())) / sigma.size
    assert abs(float(mean_product.mean()) - float(ref.mean())) <= 3.0 * image_sigma
