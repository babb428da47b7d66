#!/usr/bin/env python3
"""Derive the default-scene geometry fixture from the reference's model asset.

Reads  <reference>/assets/models/rabbit.obj  (the model main.cpp:102 loads; positions only, triangular
faces only — exactly what load_model(), main.cpp:12-62, keeps) and writes the de-indexed vertex
positions as raw little-endian float32 [n_tris, 3, 3] to rvpt_amd/assets/default_scene_tris.f32.
The output is input DATA for the default workload (BASELINE.json configs[0..1]); no reference source
text is copied.  Run in the authoring container (the reference tree does not exist on the GPU box).
"""
import sys
from pathlib import Path

import numpy as np

ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
repo = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(repo))
from rvpt_amd.scene import load_obj_positions  # noqa: E402

tris = load_obj_positions(ref / "assets" / "models" / "rabbit.obj")
assert tris.shape == (143, 3, 3), tris.shape
out = repo / "rvpt_amd" / "assets" / "default_scene_tris.f32"
tris.astype("<f4").tofile(out)
print(f"wrote {out} ({tris.shape[0]} triangles, bbox {tris.reshape(-1,3).min(0)} .. {tris.reshape(-1,3).max(0)})")
