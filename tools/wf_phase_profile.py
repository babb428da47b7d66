#!/usr/bin/env python3
"""Where the wavefront traverse kernel's time goes (GPU box).  Builds an instrumented copy of the library (-DRV_BVH_PROFILE) under
gpurun_out/, renders a few batched launches with RVPT_HIP_TIMELINE set and prints the per-phase shares of wave time and the mean
number of lanes doing useful work in each phase — the same quantities tools/bvh_phase_profile.py prints for the megakernel.
usage: [AA=4 BATCH=8] wf_phase_profile.py scene traversal"""
import os, subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
scene_name, trav = sys.argv[1], sys.argv[2]
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
lib = out / "librvpt_hip_prof.so"
if os.environ.get("RVPT_HIP_LIB") != str(lib):
    from rvpt_amd import build
    subprocess.run([build.hipcc(), *build.FLAGS, "-DRV_BVH_PROFILE", *os.environ.get("EXTRA_DEFS", "").split(), *map(str, build.SOURCES), "-o", str(lib)], check=True)
    env = dict(os.environ, RVPT_HIP_LIB=str(lib), RVPT_HIP_TIMELINE=str(out / "wf_timeline.bin"), RVPT_HIP_FRAMES_IN_FLIGHT="3")
    sys.exit(subprocess.run([sys.executable, __file__, *sys.argv[1:]], env=env).returncode)
from rvpt_amd import RVPT, native, scene  # noqa: E402
tris, mats = {"default": scene.default_scene, "cornell": scene.cornell_scene, "heightfield": scene.heightfield_scene}[scene_name]()
r = RVPT(1920, 1080, traversal=trav, flags=native.BVH_WAVEFRONT | native.COUNT_SEGMENTS)
r.add_triangles(tris)
for m in mats:
    r.add_material(m)
if scene_name == "cornell":  # bench.py's cameras
    r.scene_camera.translation = np.array([0.0, 2.0, -1.9])
elif scene_name == "heightfield":
    r.scene_camera.translation = np.array([0.0, 2.5, -5.0])
    r.scene_camera.rotation = np.array([0.0, 25.0, 0.0])
r.render_settings.aa = int(os.environ.get("AA", "4"))
r.initialize()
batch = int(os.environ.get("BATCH", "8"))
n_launches = int(os.environ.get("LAUNCHES", "4"))  # 1 = the sequence runs alone on the machine
import time
t0 = time.perf_counter()
for _ in range(n_launches):
    r.update()
    r.draw() if batch == 1 else r.draw_frames(batch)
r.wait()
wall = time.perf_counter() - t0
segments, samples = r.context.stats()
r.shutdown()  # dumps the last launch's timeline (all its traverse launches added up)
raw = np.fromfile(out / "wf_timeline.bin", dtype=np.uint64).reshape(-1, 8)
raw = raw[: len(raw) // 2]
raw = raw[raw[:, 7] > 0]
lo32 = lambda v: (v & np.uint64(0xFFFFFFFF)).astype(np.float64)
hi32 = lambda v: (v >> np.uint64(32)).astype(np.float64)
t_refill, t_inner, t_leaf, total = (raw[:, i].astype(np.float64) for i in (0, 1, 2, 7))
iters, leaf_ph = lo32(raw[:, 3]), hi32(raw[:, 3])
inner_lanes, leaf_lanes = lo32(raw[:, 4]), hi32(raw[:, 4])
dry = raw[:, 5].astype(np.float64)
refill_lanes, refills = lo32(raw[:, 6]), hi32(raw[:, 6])
tot = total.sum()
rays = segments / n_launches  # the last of the launches
print(f'wall {wall*1e3:.1f} ms for {n_launches} launch(es) of {batch} frames (includes first-use allocations)')
print(f"{scene_name} {trav} wavefront traverse: waves {len(raw)}, rays of the launch {rays:.4g} ({rays / len(raw):.0f} per wave)")
print(f"  share of wave time: refill {t_refill.sum()/tot:.3f}  inner {t_inner.sum()/tot:.3f}  leaf {t_leaf.sum()/tot:.3f}  other {1-(t_refill.sum()+t_inner.sum()+t_leaf.sum())/tot:.3f}")
print(f"  inner iterations/wave {iters.mean():.0f} ({iters.sum() * 64 / rays:.1f} lane-slots per ray), lanes walking per iteration {inner_lanes.sum()/iters.sum():.1f}; cycles per iteration {t_inner.sum()/iters.sum():.0f}; after the stream ran dry: {dry.sum()/iters.sum():.3f}")
print(f"  node visits per ray {inner_lanes.sum() / rays:.1f}")
print(f"  leaf phases/wave {leaf_ph.mean():.0f}, lanes per leaf phase {leaf_lanes.sum()/max(1,leaf_ph.sum()):.1f}; cycles per leaf phase {t_leaf.sum()/max(1,leaf_ph.sum()):.0f}; leaf visits per ray {leaf_lanes.sum() / rays:.1f}")
print(f"  refills/wave {refills.mean():.0f}, lanes refilled {refill_lanes.sum()/refills.sum():.1f}; cycles per refill {t_refill.sum()/refills.sum():.0f}")
