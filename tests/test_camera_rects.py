"""The screen rectangles of the packet kernel's camera rounds (rvpt_amd/csrc/rvpt_rect.h through rvpt_camera_rects): GPU-free checks of the host function —
every exact hit of a camera ray lies inside the hit triangle's rectangle, the fallbacks fall back, the cull is worth having.  The float-test side of the
claim (accepted by the kernels' arithmetic => inside) runs on the device: tests/test_gpu_parity.py::test_camera_rects_never_exclude_an_accepted_hit."""
import math

import numpy as np
import pytest

from _util import identity_camera, scene_by_name


def prepared_records(tris):
    """float32[n, 16] in the device's record layout (rvpt_device.h: q0 = (v0, n.x) q1 = (n.yz, e0.xy) q2 = (e0.z, e1) q3 = Gram terms; the rectangles read
    q0..q2 only): float32 edges and cross product as prepare_triangles computes them."""
    t = np.asarray(tris, np.float32).reshape(-1, 4, 4)
    v0, v1, v2 = t[:, 0, :3], t[:, 1, :3], t[:, 2, :3]
    e0, e1 = (v1 - v0).astype(np.float32), (v2 - v0).astype(np.float32)
    n = np.stack([e0[:, 1] * e1[:, 2] - e0[:, 2] * e1[:, 1], e0[:, 2] * e1[:, 0] - e0[:, 0] * e1[:, 2], e0[:, 0] * e1[:, 1] - e0[:, 1] * e1[:, 0]], 1).astype(np.float32)
    out = np.zeros((t.shape[0], 16), np.float32)
    out[:, 0:3], out[:, 3:6], out[:, 6:9], out[:, 9:12] = v0, n, e0, e1
    return out


def exact_hits(prep, cam, W, H, px, py):
    """For sample positions (px, py) in continuous pixel coordinates: bool[n_samples, n_tris], float64 Moeller-Trumbore on the records' (v0, e0, e1)
    with the ray of compute_pass.comp:151-156 + camera.glsl:29-51."""
    M = cam[:16].astype(np.float64).reshape(4, 4).T
    c0, c1, c2, o = M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3]
    aspect, w = float(cam[16]), 1.0 / math.tan(0.5 * float(cam[17]))
    cx, cy = px / W, 1.0 - py / H
    u, v = aspect * (2 * cx - 1), 2 * cy - 1
    d = u[:, None] * c0 + v[:, None] * c1 + w * c2  # (S, 3), not normalised: hits do not care
    v0, e0, e1 = prep[:, 0:3].astype(np.float64), prep[:, 6:9].astype(np.float64), prep[:, 9:12].astype(np.float64)
    hit = np.zeros((len(px), prep.shape[0]), bool)
    for j in range(prep.shape[0]):
        pv = np.cross(d, e1[j])
        det = pv @ e0[j]
        with np.errstate(divide="ignore", invalid="ignore"):
            tv = o - v0[j]
            uu = (pv @ tv) / det
            qv = np.cross(tv, e0[j])
            vv = (d @ qv) / det
            tt = (qv @ e1[j]) / det
        hit[:, j] = (uu > 0) & (vv > 0) & (uu + vv < 1) & (tt > 0)
    return hit


CAMERAS = [("default", (0, 0, 0), (0, 0, 0), 90.0), ("framing", (0, 0.9, -2.5), (0, 0, 0), 90.0), ("oblique", (1.4, 1.6, -1.2), (-40.0, 25.0, 10.0), 70.0),
           ("inside", (-0.1, 0.8, 0.05), (120.0, -10.0, 0.0), 110.0)]


@pytest.mark.parametrize("scene_name", ["default", "showcase"])
@pytest.mark.parametrize("cam_name,tr,rot,fov", CAMERAS)
@pytest.mark.parametrize("W,H", [(1920, 1080), (208, 120)])
def test_every_exact_hit_lies_inside_the_triangles_rectangle(scene_name, cam_name, tr, rot, fov, W, H):
    from rvpt_amd import Camera, native
    tris, _, _ = scene_by_name(scene_name)
    prep = prepared_records(tris)
    c = Camera(W / H)
    c.translation, c.rotation, c.fov = np.array(tr, float), np.array(rot, float), fov
    cam = c.get_data()
    rects = native.camera_rects(prep, cam, W, H)
    rng = np.random.default_rng(7)
    S = 6000
    px, py = rng.uniform(0, W, S), rng.uniform(0, H, S)
    # samples on block borders too: the rectangle's outward rounding is what they test
    px[:500] = np.round(px[:500] / 16) * 16
    py[500:1000] = np.round(py[500:1000] / 4) * 4
    px, py = np.clip(px, 0, W - 1e-9), np.clip(py, 0, H - 1e-9)
    hit = exact_hits(prep, cam, W, H, px, py)
    bx, by = (px // 16).astype(int)[:, None], (py // 4).astype(int)[:, None]
    inside = (bx >= rects[None, :, 0]) & (bx <= rects[None, :, 1]) & (by >= rects[None, :, 2]) & (by <= rects[None, :, 3])
    assert hit.any() or cam_name == "inside"
    assert not (hit & ~inside).any(), np.argwhere(hit & ~inside)[:5]


def test_the_cull_is_worth_having_on_the_headline_frame():
    """Default scene, default camera, 1920 x 1080: the rectangles leave < 3 % of all (16 x 4 block, triangle) pairs (1.7 % as built); the 22 triangles
    behind the camera plane are out altogether."""
    from rvpt_amd import Camera, native
    tris, _, _ = scene_by_name("default")
    W, H = 1920, 1080
    r = native.camera_rects(prepared_records(tris), Camera(W / H).get_data(), W, H).astype(np.int64)
    nbx, nby = W // 16, H // 4
    x0, x1, y0, y1 = np.clip(r[:, 0], 0, nbx), np.clip(r[:, 1], -1, nbx - 1), np.clip(r[:, 2], 0, nby), np.clip(r[:, 3], -1, nby - 1)
    held = (np.maximum(0, x1 - x0 + 1) * np.maximum(0, y1 - y0 + 1)).sum()
    assert held / (len(r) * nbx * nby) < 0.03
    assert ((r[:, 0] > r[:, 1]) | (r[:, 2] > r[:, 3])).sum() >= 20


def test_whatever_breaks_a_premise_gives_the_whole_screen():
    from rvpt_amd import native
    W, H = 640, 360
    whole = np.array([0, 65535, 0, 65535])
    tri = np.zeros((1, 4, 4), np.float32)
    tri[0, 0, :3], tri[0, 1, :3], tri[0, 2, :3] = (-1, -1, 3), (1, -1, 3), (0, 1, 3)
    ok = native.camera_rects(prepared_records(tri), identity_camera(W / H), W, H)[0]
    # x in [-1, 1] at depth 3, 90 degrees: pixels 320 -+ 60 -> blocks 16 .. 23, one block of margin; y: pixels 180 -+ 60 -> 4-row blocks 30 .. 60
    assert ok[0] in (14, 15) and ok[1] in (24, 25) and ok[2] in (28, 29) and ok[3] in (61, 62), ok
    # the camera in the triangle's plane (the record is not `safe`: numerator 0)
    flat = tri.copy()
    flat[0, :, 2] = 0.0
    assert (native.camera_rects(prepared_records(flat), identity_camera(W / H), W, H)[0] == whole).all()
    # a sliver (sin^2 of the angle between the edges below 2^-12), absurd scales, NaN in the record, NaN / singular / absurd cameras
    sliver = tri.copy()
    sliver[0, 2, :3] = sliver[0, 0, :3] + (sliver[0, 1, :3] - sliver[0, 0, :3]) * 0.5 + np.array([0, 1e-3, 0], np.float32)
    assert (native.camera_rects(prepared_records(sliver), identity_camera(W / H), W, H)[0] == whole).all()
    huge = prepared_records(tri * np.float32(1e25))
    assert (native.camera_rects(huge, identity_camera(W / H), W, H)[0] == whole).all()
    nan = prepared_records(tri)
    nan[0, 7] = np.nan
    assert (native.camera_rects(nan, identity_camera(W / H), W, H)[0] == whole).all()
    for bad in ("nan", "singular", "fov", "aspect"):
        cam = identity_camera(W / H)
        if bad == "nan":
            cam[13] = np.nan
        elif bad == "singular":
            cam[0:3] = cam[4:7]
        elif bad == "fov":
            cam[17] = math.radians(179.9)
        else:
            cam[16] = 0.0
        assert (native.camera_rects(prepared_records(tri), cam, W, H)[0] == whole).all(), bad
    # behind the camera plane: no block; crossing it: runs off to the screen edge on the side the projected edges move towards
    behind = tri.copy()
    behind[0, :, 2] = -3.0
    r = native.camera_rects(prepared_records(behind), identity_camera(W / H), W, H)[0]
    assert r[0] > r[1]
    cross = tri.copy()
    cross[0, 0, :3], cross[0, 1, :3], cross[0, 2, :3] = (0.2, -0.2, 2.0), (0.2, 0.2, 2.0), (3.0, 0.0, -1.0)  # tip behind the camera, to the right
    r = native.camera_rects(prepared_records(cross), identity_camera(W / H), W, H)[0]
    assert r[1] == 65535 and r[0] == 20  # from just right of the centre (pixel 338 = block 21, one block of margin) to +infinity in x
    # a mirrored (left-handed) camera matrix is still a camera: same hits, mirrored rectangle
    cam = identity_camera(W / H)
    cam[0] = -1.0
    r = native.camera_rects(prepared_records(tri), cam, W, H)[0]
    assert (r != whole).any()
