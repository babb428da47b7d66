"""The bounce cull's table (rvpt_amd/csrc/rvpt_vis.h through rvpt_bounce_rows of the laboratory build — the function upload_scene runs on the device): GPU-free
checks of its rows against half-space tests evaluated exactly (rationals) and in float64 with a slack band around every threshold.  A cleared bit must be
PROVABLE: triangle B wholly behind the plane of A as seen from the row's side by the margin, both triangles well shaped; and the table must be worth having:
what is clearly behind is out.  The float-test side of the claim (accepted on a segment that leaves A => in A's row) runs on the device:
tests/test_gpu_parity.py::test_bounce_cull_never_excludes_an_accepted_hit and tools/fuzz_culls.py."""
from fractions import Fraction

import numpy as np
import pytest

from _util import scene_by_name
from test_camera_rects import prepared_records


def scene_scale(tris):
    t = np.asarray(tris, np.float64).reshape(-1, 4, 4)[:, :3, :3].reshape(-1, 3)
    return float(np.abs(t).max() + (t.max(0) - t.min(0)).max())


def float64_rows(prep, margin, slack):
    """(must_be_set, must_be_clear) boolean [2 n, n]: outside the slack band around the thresholds the double evaluation cannot differ from the exact one."""
    p = prep.astype(np.float64)
    v0, n, e0, e1 = p[:, 0:3], p[:, 3:6], p[:, 6:9], p[:, 9:12]
    a00, a11, a01 = (e1 * e1).sum(1), (e0 * e0).sum(1), (e0 * e1).sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        kappa = (a00 * a11 - a01 * a01) / (a00 * a11)
    nn = np.sqrt((n * n).sum(1))
    ok_hi = (kappa >= 2.0 ** -6 * (1 + slack)) & (a00 * a11 > 0) & (nn > 0)     # certainly well shaped
    ok_lo = (kappa >= 2.0 ** -6 * (1 - slack)) & (a00 * a11 > 0) & (nn > 0)     # possibly well shaped
    verts = np.stack([v0, v0 + e0, v0 + e1], 1)  # [n, 3, 3]
    N = len(p)
    must_set = np.zeros((2 * N, N), bool)
    must_clear = np.zeros((2 * N, N), bool)
    for A in range(N):
        with np.errstate(invalid="ignore", divide="ignore"):
            d = ((verts - v0[A]) @ n[A]) / nn[A]  # [n, 3] signed distances to A's plane
        for s, sign in ((0, 1.0), (1, -1.0)):
            behind_hi = (sign * d <= -margin * (1 + slack)).all(1)
            behind_lo = (sign * d <= -margin * (1 - slack)).all(1)
            must_clear[2 * A + s] = behind_hi & ok_hi & ok_hi[A]
            must_set[2 * A + s] = ~(behind_lo & ok_lo & ok_lo[A])
    return must_set, must_clear


def unpack(rows, n):
    bits = ((rows[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(rows.shape[0], -1).astype(bool)
    assert not bits[:, n:].any(), "bits beyond the last triangle must be 0"
    return bits[:, :n]


def soup(rng, n, decades=3.0):
    c = rng.uniform(-1, 1, (n, 1, 3))
    return (c + 10.0 ** rng.uniform(-decades, 0, (n, 1, 1)) * rng.normal(size=(n, 3, 3))).astype(np.float32)


def triangles(pos):
    from rvpt_amd import scene
    return scene.make_triangles(pos, 0)


@pytest.mark.parametrize("name", ["default", "showcase", "soup", "slivers", "scaled_up", "scaled_down", "translated", "box"])
def test_rows_against_float64_half_spaces(name):
    from rvpt_amd import native
    rng = np.random.default_rng(11)
    if name in ("default", "showcase"):
        tris = scene_by_name(name)[0]
    elif name == "soup":
        tris = triangles(soup(rng, 300))
    elif name == "slivers":  # edge angles swept through the 2^-6 threshold
        pos = soup(rng, 120, 1.0).astype(np.float64)
        for i in range(0, 120, 2):
            e0 = pos[i, 1] - pos[i, 0]
            perp = np.cross(e0, rng.normal(size=3))
            perp /= np.linalg.norm(perp)
            kappa = 2.0 ** -6 * float(rng.choice([0.25, 0.5, 0.999, 1.001, 2.0, 4.0]))
            th = np.arcsin(np.sqrt(kappa))
            pos[i, 2] = pos[i, 0] + np.linalg.norm(e0) * (np.cos(th) * e0 / np.linalg.norm(e0) + np.sin(th) * perp)
        tris = triangles(pos.astype(np.float32))
    elif name == "scaled_up":
        tris = triangles(soup(rng, 150) * np.float32(2.0 ** 20))
    elif name == "scaled_down":
        tris = triangles(soup(rng, 150) * np.float32(2.0 ** -20))
    elif name == "translated":
        tris = triangles(soup(rng, 150) + np.float32(100.0))
    else:  # twelve large triangles around small geometry
        import sys
        sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "tools"))
        import fuzz_culls
        inner = soup(rng, 60, 2.0).astype(np.float64) * 0.1
        tris = triangles(np.concatenate([fuzz_culls._box(3.0, np.zeros(3)), inner]).astype(np.float32))
    prep = prepared_records(tris)
    rows, scale = native.bounce_rows(tris, prep)
    n = len(tris)
    assert rows.shape == (2 * n, (n + 31) // 32)
    assert scale == pytest.approx(scene_scale(tris), rel=1e-12)
    bits = unpack(rows, n)
    must_set, must_clear = float64_rows(prep, 2.0 ** -10 * scale, 1e-9)
    assert not (must_set & ~bits).any(), np.argwhere(must_set & ~bits)[:5]      # soundness: nothing is culled that is not provably behind
    assert not (must_clear & bits).any(), np.argwhere(must_clear & bits)[:5]    # ... and what is clearly behind IS culled
    # a triangle is never behind its own plane
    assert all(bits[2 * a, a] and bits[2 * a + 1, a] for a in range(n))
    if name in ("default", "box", "soup"):
        assert bits.mean() < 0.85  # worth having (the default scene: 0.62)


def test_rows_against_exact_rationals():
    """Forty triangles, every one of the 2 n^2 bits decided in exact rational arithmetic on the float records: a cleared bit has all three vertices of B at
    signed distance <= -margin (1 - 2^-40) from A's plane on the row's side and both kappa >= 2^-6 (1 - 2^-40)."""
    from rvpt_amd import native
    rng = np.random.default_rng(5)
    tris = triangles(soup(rng, 40, 1.5))
    prep = prepared_records(tris)
    rows, scale = native.bounce_rows(tris, prep)
    n = len(tris)
    bits = unpack(rows, n)
    F = lambda x: Fraction(float(x))
    margin = F(2.0 ** -10 * scale) * (1 - Fraction(1, 2 ** 40))
    rec = [[F(x) for x in row] for row in prep]

    def dot(a, b):
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    def kappa_ok(r):
        e0, e1 = r[6:9], r[9:12]
        a00, a11, a01 = dot(e1, e1), dot(e0, e0), dot(e0, e1)
        return a00 * a11 > 0 and (a00 * a11 - a01 * a01) >= Fraction(1, 64) * (1 - Fraction(1, 2 ** 40)) * a00 * a11

    cleared = 0
    for A in range(n):
        v0a, na = rec[A][0:3], rec[A][3:6]
        nn2 = dot(na, na)
        for s, sign in ((0, 1), (1, -1)):
            for B in range(n):
                if bits[2 * A + s, B]:
                    continue
                cleared += 1
                assert kappa_ok(rec[A]) and kappa_ok(rec[B]) and nn2 > 0
                v0, e0, e1 = rec[B][0:3], rec[B][6:9], rec[B][9:12]
                for vert in (v0, [v0[k] + e0[k] for k in range(3)], [v0[k] + e1[k] for k in range(3)]):
                    d = sign * dot([vert[k] - v0a[k] for k in range(3)], na)
                    assert d < 0 and d * d >= margin * margin * nn2, (A, s, B)  # d / |n| <= -margin
    assert cleared > 100


def test_no_table_for_scenes_without_a_scale():
    from rvpt_amd import native
    rng = np.random.default_rng(3)
    tris = triangles(soup(rng, 8))
    bad = tris.copy()
    bad[3, 5] = np.nan
    assert native.bounce_rows(bad, prepared_records(bad))[1] == 0.0
    huge = tris.copy()
    huge[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] *= np.float32(1e30)
    with np.errstate(over="ignore", invalid="ignore"):
        assert native.bounce_rows(huge, prepared_records(huge))[1] == 0.0
    # a degenerate triangle (zero area) culls nothing and is culled by nothing
    deg = tris.copy()
    deg[2, 8:11] = deg[2, 4:7]
    rows, scale = native.bounce_rows(deg, prepared_records(deg))
    bits = unpack(rows, len(deg))
    assert scale > 0 and bits[4].all() and bits[5].all() and bits[:, 2].all()


def test_leaf_boxes_hold_their_triangles_with_the_margin():
    """The leaf boxes of the bounce rounds (rvpt_vis.h: bounce_leaf_boxes through rvpt_bounce_leaf_boxes): every group of consecutive triangles lies inside its box
    with at least the margin 2^-9 (scale + 2 EPSILON) to spare on every side (float32 rounding of the bounds: 2^-24 of a coordinate), the boxes are not absurdly loose,
    and a group with a sliver, a zero-area or a non-finite triangle gets the infinite box."""
    from rvpt_amd import native
    rng = np.random.default_rng(9)
    for name in ("default", "soup", "scaled_down"):
        tris = scene_by_name("default")[0] if name == "default" else triangles(soup(rng, 203) * np.float32(1.0 if name == "soup" else 2.0 ** -18))
        boxes, per = native.bounce_leaf_boxes(tris)
        assert per in (4, 8) and boxes.shape == ((len(tris) + per - 1) // per, 8)
        M = 2.0 ** -9 * (scene_scale(tris) + 0.01)
        v = np.asarray(tris, np.float64).reshape(-1, 4, 4)[:, :3, :3]
        kappa_ok = float64_rows(prepared_records(tris), 1.0, 0.0)  # (only to have the records' kappa: recomputed below)
        p = prepared_records(tris).astype(np.float64)
        e0, e1 = p[:, 6:9], p[:, 9:12]
        a00, a11, a01 = (e1 * e1).sum(1), (e0 * e0).sum(1), (e0 * e1).sum(1)
        with np.errstate(invalid="ignore", divide="ignore"):
            good = ((a00 * a11 - a01 * a01) >= 2.0 ** -6 * a00 * a11) & (a00 * a11 > 0)
        for l in range(len(boxes)):
            t = v[l * per:(l + 1) * per].reshape(-1, 3)
            lo, hi = boxes[l, 0:3].astype(np.float64), boxes[l, 3:6].astype(np.float64)
            if not good[l * per:(l + 1) * per].all():
                assert np.isneginf(lo).all() and np.isposinf(hi).all()
                continue
            slack = 2.0 ** -22 * (np.abs(t).max() + M)
            assert (t.min(0) - lo >= M - slack).all() and (hi - t.max(0) >= M - slack).all(), (name, l)
            assert (t.min(0) - lo <= 1.01 * M + slack).all() and (hi - t.max(0) <= 1.01 * M + slack).all(), (name, l)
    bad = triangles(soup(rng, 16))
    bad[5, 8:11] = bad[5, 4:7]     # zero area
    bad[11, 1] = np.inf
    boxes, per = native.bounce_leaf_boxes(bad)
    if np.isfinite(bad[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]]).all():
        assert np.isinf(boxes[5 // per, :6]).all()
    else:
        assert (boxes == 0).all()  # a non-finite coordinate anywhere: the scene has no scale, no table, no boxes
