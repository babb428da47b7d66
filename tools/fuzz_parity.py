#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random image sizes, spp, bounce budgets, per-quadrant integrators, split ratios,
camera modes and poses, traversals, kernels (regenerating / simple), frames per launch, tile partitions and scenes — every case must match
the oracle bit for bit.   python tools/fuzz_parity.py [n_cases] [seed]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402
from rvpt_amd import Camera, RenderSettings, native, scene  # noqa: E402



def run(n_cases: int, seed: int) -> int:
    """Returns the number of mismatching cases (prints each)."""
    rng = np.random.RandomState(seed)
    scenes = {}
    for name, make in (("default", scene.default_scene), ("showcase", scene.materials_showcase_scene),
                       ("terrain", lambda: scene.heightfield_scene(cells=30)), ("model2k", lambda: (scene.make_triangles(scene.subdivide(scene.default_model_positions(), 2), 1), scene.default_materials()))):
        tris, mats = make()
        nodes, idx = native.build_bvh(tris)
        scenes[name] = (tris[idx], mats, nodes)

    import os
    only = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None
    bad = 0
    for case in range(n_cases):
        sname = rng.choice(list(scenes))
        tris, mats, nodes = scenes[sname]
        W, H = int(rng.randint(1, 97)), int(rng.randint(1, 70))
        if os.environ.get("FUZZ_BIG") and sname in ("default", "showcase") and rng.rand() < 0.5:
            # images large enough for the packet kernel's aligned camera rounds (screen rectangles) and cost-relevant tile counts
            W, H = int(rng.randint(97, 520)), int(rng.randint(70, 300))
        trav = rng.choice(["brute", "bvh", "bvh_ordered"])
        world = int(rng.choice([1, 1, 2, 3]))
        simple = bool(rng.rand() < 0.2)
        modes = [int(rng.choice([9, 9, 9, 0, 1, 2, 3, 4, 5, 6, 7, 8, 11])) for _ in range(4)] if rng.rand() < 0.6 else [9] * 4
        if sname in ("terrain", "model2k") and trav == "brute" and any(m > 9 for m in modes):
            modes = [m if m <= 9 else 9 for m in modes]  # keep the O(32 N) heat map to the small scenes
        kw = dict(max_bounces=int(rng.randint(0, 7)), aa=int(rng.randint(1, 4)), camera_mode=int(rng.choice([0, 0, 0, 1, 2])),
                  split=(float(np.float32(rng.rand())), float(np.float32(rng.rand()))))
        c = Camera(W / H)
        c.translation = rng.uniform(-1.5, 1.5, 3) + np.array([0, 1.0, -2.0])
        c.rotation = rng.uniform(-25, 25, 3)
        c.set_fov(float(rng.uniform(40, 110)))
        cam = c.get_data()
        frames = int(rng.randint(1, 5))
        batched = bool(rng.rand() < 0.5)
        unorm8 = bool(rng.rand() < 0.15)       # the reference's storage: the running mean re-quantised to rgba8 every frame
        collective = bool(rng.rand() < 0.15)   # read through the library's RCCL communicator (world 1: a self send / recv)
        flags = {"bvh": native.TRAVERSAL_BVH, "brute": 0, "bvh_ordered": native.TRAVERSAL_BVH_ORDERED}[trav] | (native.KERNEL_SIMPLE if simple else 0) | native.COUNT_SEGMENTS
        if unorm8:
            flags |= native.ACCUM_UNORM8
        got = np.zeros((H, W, 4), np.float32)
        seg_gpu = 0
        plans = []  # per rank: frames per dispatch, drawn before anything runs so that FUZZ_ONLY=<case> replays one case exactly
        for rank in range(world):
            plan, f = [], 0
            while f < frames:  # frame by frame, or a random batch of consecutive frames as one launch
                n = 1 if not batched else int(rng.randint(1, frames - f + 1))
                plan.append(n)
                f += n
            plans.append(plan)
        if only is not None and case != only:
            continue
        if only is not None:
            print(f"case {case}: {sname} {W}x{H} {trav} world={world} simple={simple} unorm8={unorm8} collective={collective} plans={plans} modes={modes} {kw} frames={frames}")
        for rank in range(world):
            ctx = native.Context(W, H, 0, rank, world, flags)
            ctx.upload_scene(nodes if trav != "brute" else None, tris, mats)
            if collective and world == 1:
                ctx.comm_init(native.comm_unique_id())
            f = 0
            for n in plans[rank]:
                rs = RenderSettings(max_bounces=kw["max_bounces"], aa=kw["aa"], current_frame=f, camera_mode=kw["camera_mode"],
                                    top_left_render_mode=modes[0], top_right_render_mode=modes[1], bottom_left_render_mode=modes[2],
                                    bottom_right_render_mode=modes[3], split_ratio=kw["split"])
                ctx.set_frame(rs.pack(), cam)
                ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
                f += n
            got += ctx.read()
            seg_gpu += ctx.stats()[0]
            ctx.close()
        prev, seg = None, 0
        for f in range(frames):
            prev, st = oracle.render(oracle.settings_bytes(current_frame=f, modes=tuple(modes), **kw), cam, nodes, tris, mats, W, H,
                                     {"bvh": oracle.TRAVERSAL_BVH, "brute": oracle.TRAVERSAL_BRUTE, "bvh_ordered": oracle.TRAVERSAL_BVH_ORDERED}[trav], prev=prev)
            if unorm8:
                prev = oracle.dequantize_rgba8(oracle.quantize_rgba8(prev))
            seg += int(st[0])
        same = np.array_equal(np.nan_to_num(got, nan=-7.0).view(np.uint32), np.nan_to_num(prev, nan=-7.0).view(np.uint32)) and np.array_equal(np.isnan(got), np.isnan(prev))
        if not same or seg != seg_gpu:
            bad += 1
            print(f"MISMATCH case {case}: {sname} {W}x{H} {trav} world={world} simple={simple} batched={batched} unorm8={unorm8} collective={collective} modes={modes} {kw} frames={frames} "
                  f"pixels differing={int((got != prev).any(axis=2).sum())} segments {seg_gpu} vs {seg}")
    print(f"{n_cases - bad}/{n_cases} cases bit-identical")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
