"""The oracle against the committed fixtures (tests/golden/, made by tools/make_golden.py).

Upstream has no golden vectors; these pin the oracle's own output bit-for-bit so that it cannot drift
silently underneath the GPU parity tests."""
import numpy as np
import pytest

from _util import GOLDEN, scene_by_name

CASES = sorted(p.stem for p in GOLDEN.glob("*.npz"))


def scene_for(name):
    return scene_by_name(name.split("_")[0])


def test_fixture_set_is_complete():
    assert len(CASES) == 14


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_fixture(oracle, case):
    fx = np.load(GOLDEN / f"{case}.npz")
    tris, mats, nodes = scene_for(case)
    trav = {"brute": oracle.TRAVERSAL_BRUTE, "bvh": oracle.TRAVERSAL_BVH, "bvhordered": oracle.TRAVERSAL_BVH_ORDERED}[case.split("_")[-1]]
    W, H = int(fx["width"]), int(fx["height"])
    prev = None
    for f in range(4):
        s = oracle.settings_bytes(max_bounces=int(fx["max_bounces"]), aa=int(fx["aa"]), current_frame=f)
        img, stats = oracle.render(s, fx["camera"], nodes, tris, mats, W, H, trav, prev=prev)
        prev = img
        if f in (0, 3):
            assert np.array_equal(img, fx[f"frame{f}"]), f"{case} frame {f} drifted"
            assert stats.tolist() == fx[f"stats{f}"].tolist()
