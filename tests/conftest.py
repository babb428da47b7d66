import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure) — compiled on demand with gcc."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def default_scene():
    """(sorted_tris, mats, nodes) of the reference default scene in BVH-leaf order."""
    from rvpt_amd import native, scene
    tris, mats = scene.default_scene()
    nodes, idx = native.build_bvh(tris)
    return tris[idx], mats, nodes
