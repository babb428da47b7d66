#!/usr/bin/env python3
"""One dispatch per frame — the reference's own shape (RVPT::draw submits one compute pass per frame, rvpt.cpp:346-354; a moving camera leaves nothing to batch) — on
the headline frame (default scene, 1920x1080, 1 spp, 8 bounces, brute force): Msamples/s of K frames sent out one launch each, no wait in between, wall clock from the
first set_frame to the drain.   usage: tools/one_frame_per_launch.py [--frames K] [--timing] [--moving]     (environment knobs apply: RVPT_HIP_FRAMES_IN_FLIGHT, and
with RVPT_HIP_LAB=1 the laboratory build's RVPT_HIP_BLOCKS_PER_CU ...)"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402

from rvpt_amd import RVPT, native, scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--timing", action="store_true", help="bracket every frame kernel with hipEvents, as bench.py does")
ap.add_argument("--moving", action="store_true", help="move the camera every frame (the accumulation restarts: current_frame = 0 every time, new rectangles every launch)")
ap.add_argument("--count", action="store_true", help="count segments and samples (RVPT_HIP_COUNT_SEGMENTS), as bench.py does")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
W, H = 1920, 1080
tris, mats = scene.default_scene()
r = RVPT(W, H, device=0, traversal="brute", flags=(native.TIMING if a.timing else 0) | (native.COUNT_SEGMENTS if a.count else 0))
r.add_triangles(tris)
for m in mats:
    r.add_material(m)
r.initialize()


def run(n):
    for i in range(n):
        if a.moving:
            r.scene_camera.translate((0.0, 0.0, 1e-3))
        r.update()
        r.draw()


run(300)
r.wait()
best = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    run(a.frames)
    t_host = time.perf_counter() - t0
    r.wait()
    dt = time.perf_counter() - t0
    best.append((dt, t_host))
dt, t_host = sorted(best)[len(best) // 2]
print(f"one frame per launch, {a.frames} frames{', hipEvent timing' if a.timing else ''}{', counting segments' if a.count else ''}{', moving camera' if a.moving else ''}, in flight {r.context.launch_info()[3]}: "
      f"{W * H * a.frames / dt / 1e6:.0f} Msamples/s, {dt / a.frames * 1e6:.1f} us per frame (host enqueue {t_host / a.frames * 1e6:.1f} us per frame)")
r.shutdown()
