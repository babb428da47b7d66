#!/usr/bin/env python3
"""Where the BVH kernel's time goes (GPU box).  Builds an instrumented copy of the library (-DRV_BVH_PROFILE) under
gpurun_out/, renders a few frames with RVPT_HIP_TIMELINE set and prints the per-phase shares of wave time and the
mean number of lanes doing useful work in each phase.  usage: bvh_phase_profile.py scene traversal [frames-in-flight]"""
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
    cmd = [build.hipcc(), *build.FLAGS, "-DRVPT_HIP_LAB=1", "-DRV_BVH_PROFILE", *map(str, build.LAB_SOURCES), "-o", str(lib)]  # (RVPT_HIP_TIMELINE is a knob of the laboratory build)
    subprocess.run(cmd, check=True)
    env = dict(os.environ, RVPT_HIP_LIB=str(lib), RVPT_HIP_TIMELINE=str(out / "bvh_timeline.bin"), RVPT_HIP_FRAMES_IN_FLIGHT=sys.argv[3] if len(sys.argv) > 3 else "3")
    sys.exit(subprocess.run([sys.executable, __file__, *sys.argv[1:]], env=env).returncode)
from rvpt_amd import RVPT, scene  # noqa: E402
tris, mats = {"default": scene.default_scene, "cornell": scene.cornell_scene, "heightfield": scene.heightfield_scene}[scene_name]()
# WIDE=1: the default walk over the 4-wide tree (rvpt_bvh4.hip carries the same instrumentation); else rounds 1-3's binary per-lane kernel
r = RVPT(1920, 1080, traversal=trav, flags=0 if os.environ.get("WIDE") else __import__("rvpt_amd").native.BVH_PER_LANE,
         tile_rank=int(os.environ.get("RANK_OF", "0/1").split("/")[0]), tile_world=int(os.environ.get("RANK_OF", "0/1").split("/")[1]))
r.add_triangles(tris)
for m in mats:
    r.add_material(m)
if scene_name == "cornell":  # bench.py's cameras
    r.scene_camera.translation = np.array([0.0, 2.0, -1.9])
elif scene_name == "heightfield":
    r.scene_camera.translation = np.array([0.0, 2.5, -5.0])
    r.scene_camera.rotation = np.array([0.0, 25.0, 0.0])
r.render_settings.aa = int(os.environ.get("AA", "1"))
r.initialize()
batch = int(os.environ.get("BATCH", "1"))
for _ in range(6):
    r.update()
    r.draw() if batch == 1 else r.draw_frames(batch)
    if os.environ.get("LONE"):  # every launch alone on the chip (a rank's K-step share as one launch)
        r.wait()
r.wait()
r.shutdown()  # dumps the last frame's timeline
raw = np.fromfile(out / "bvh_timeline.bin", dtype=np.uint64).reshape(-1, 8)
extra = raw[len(raw) // 2:]  # second half: wave-level global load instructions by source (and the lanes active in them)
raw = raw[:len(raw) // 2]
extra = extra[raw[:, 7] > 0]
raw = raw[raw[:, 7] > 0]
lo32 = lambda v: (v & np.uint64(0xFFFFFFFF)).astype(np.float64)
hi32 = lambda v: (v >> np.uint64(32)).astype(np.float64)
t_refill, t_inner = (raw[:, i].astype(np.float64) for i in range(2))
t_leaf = (raw[:, 2] & np.uint64((1 << 40) - 1)).astype(np.float64)
t_pop = (raw[:, 2] >> np.uint64(40)).astype(np.float64)  # (wide kernel only: the pops' share of `inner`)
total = (raw[:, 7] & np.uint64((1 << 40) - 1)).astype(np.float64)
dry = (raw[:, 7] >> np.uint64(40)).astype(np.float64)
iters, leaf_ph = lo32(raw[:, 3]), hi32(raw[:, 3])
inner_lanes, leaf_lanes = lo32(raw[:, 4]), hi32(raw[:, 4])
hist = np.stack([((raw[:, 5] >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.float64) for k in range(4)], axis=1).sum(axis=0)
refill_lanes, refills = lo32(raw[:, 6]), hi32(raw[:, 6])
tot = total.sum()
print(f"{scene_name} {trav}: waves {len(raw)}")
print(f"  share of wave time: refill(shade+regen) {t_refill.sum()/tot:.3f}  inner {t_inner.sum()/tot:.3f}  leaf {t_leaf.sum()/tot:.3f}  other {1-(t_refill.sum()+t_inner.sum()+t_leaf.sum())/tot:.3f}")
if t_pop.sum() > 0:
    print(f"    of which pops {t_pop.sum()/tot:.3f} (cycles per iteration {t_pop.sum()/iters.sum():.0f})")
print(f"  inner iterations/wave {iters.mean():.0f}, lanes walking per iteration {inner_lanes.sum()/iters.sum():.1f}; cycles per iteration {t_inner.sum()/iters.sum():.0f}")
print(f"    iterations by lanes walking 0-16/17-32/33-48/49-64: {np.round(hist/hist.sum(),3).tolist()}; after the pixel pool ran dry: {dry.sum()/iters.sum():.3f}")
print(f"  leaf phases/wave {leaf_ph.mean():.0f}, lanes per leaf phase {leaf_lanes.sum()/max(1,leaf_ph.sum()):.1f}; cycles per leaf phase {t_leaf.sum()/max(1,leaf_ph.sum()):.0f}")
print(f"  refills/wave {refills.mean():.0f}, lanes refilled {refill_lanes.sum()/refills.sum():.1f}; cycles per refill {t_refill.sum()/refills.sum():.0f}")
if os.environ.get("WIDE"):  # the wide kernel's second half: wall-clock stamps (100 MHz) of every wave of the LAST launch — ramp / steady / tail
    if extra[:, 4].sum() > 0:  # issue-to-arrival time of a step's node loads as the wave sees it (exact 128-byte nodes)
        print(f"  node loads: {extra[:, 3].sum() / extra[:, 4].sum():.0f} cycles from issue to arrival per step ({extra[:, 4].sum() / len(extra):.0f} timed steps per wave)")
    t0, dry, t1 = (extra[:, i].astype(np.int64) for i in range(3))
    ok = t1 > 0
    t0, dry, t1 = t0[ok], dry[ok], t1[ok]
    base = t0.min()
    us = lambda x: (x - base) / 100.0
    d = dry[dry > 0]
    print(f"  last launch: span {us(t1.max()):.1f} us; waves start by {us(t0.max()):.1f} us (median {np.median(us(t0)):.1f}); pixel pool dry first {us(d.min()) if len(d) else -1:.1f} / median {np.median(us(d)) if len(d) else -1:.1f} / last {us(d.max()) if len(d) else -1:.1f} us")
    print(f"  wave end: first {us(t1.min()):.1f}, 10% {np.percentile(us(t1), 10):.1f}, median {np.median(us(t1)):.1f}, 90% {np.percentile(us(t1), 90):.1f}, last {us(t1.max()):.1f} us; mean busy share of the span {((t1 - t0).mean() / 100.0) / us(t1.max()):.3f}")
    sys.exit(0)
ld = extra.astype(np.float64).sum(axis=0)
for name, i in (("node pairs (4 x dwordx4 per step)", 0), ("popped heads (dwordx2)", 2), ("leaf triangles (4 x dwordx4 each)", 4)):
    print(f"  global loads, {name}: {ld[i]:.4g} wave-level instructions per launch, {ld[i+1]/max(1,ld[i]):.1f} lanes active")
