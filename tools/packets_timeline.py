#!/usr/bin/env python3
"""Run on the GPU box: ONE lone launch of the packet kernel over N frames with the per-wave timeline (RVPT_HIP_TIMELINE), split into ramp / steady / tail.
usage: tools/packets_timeline.py [frames=20] [emulate_world=1] [rank=0]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "build" else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.environ["RVPT_HIP_TIMELINE"] = str(out / "packets_timeline.bin")
# the instrumented build (-DRV_PACKETS_TIMELINE=1): made here if tools/packets_timeline.py build was not run before the snapshot
from rvpt_amd import build as B
lib = ROOT / "build" / "exp" / "packets_timeline.so"
if not lib.exists() or lib.stat().st_mtime < max(p.stat().st_mtime for p in B.SOURCES + B.HEADERS):
    import subprocess
    lib.parent.mkdir(parents=True, exist_ok=True)
    subprocess.run([B.hipcc(), *B.FLAGS, "-DRVPT_HIP_LAB=1", "-DRV_PACKETS_TIMELINE=1", "-DRV_PACKETS_MIN_WAVES=5",  # (five waves per SIMD: the clocks and counters need the registers of the sixth)
                     *map(str, B.LAB_SOURCES), "-o", str(lib)], check=True)  # (RVPT_HIP_TIMELINE is a knob of the laboratory build)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
os.environ["RVPT_HIP_LIB"] = str(lib)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
from rvpt_amd import RVPT, native, scene
tris, mats = scene.default_scene()
r = RVPT(1920, 1080, device=0, traversal="brute", tile_rank=rank, tile_world=world, flags=native.TIMING)
r.add_triangles(tris)
for m in mats:
    r.add_material(m)
r.initialize()
for _ in range(6):  # clocks, buffers
    r.update(); r.draw_frames(frames); r.wait()
r.context.reset_timing()
r.update(); r.draw_frames(frames); r.wait()
ms = r.context.timing()[0]
grid, r_lds = r.context.launch_info()[:2]
r.shutdown()
whole = np.fromfile(out / "packets_timeline.bin", dtype=np.uint64).reshape(-1, 8)
raw = whole[: grid * 4]
phases = whole[len(whole) // 2: len(whole) // 2 + grid * 4].astype(np.float64)  # second half of the rows: shader clocks per phase (rvpt_packets.hip: RV_PHASE)
t0, dry, t1 = raw[:, 0].astype(np.int64), raw[:, 1].astype(np.int64), raw[:, 2].astype(np.int64)
base = t0.min()
us = lambda x: (x - base) / 100.0
print(f"frames {frames} world {world} rank {rank}: kernel {ms * 1e3:.1f} us (hipEvents), waves {len(raw)}; wall clock span {us(t1.max()):.1f} us")
print(f"  LDS {r_lds} B per work-group, grid {grid}; waves that start later than 20 us: {(us(t0) > 20).sum()}")
print(f"  wave start: median {np.median(us(t0)):.1f} us, last {us(t0.max()):.1f} us")
d = dry[dry > 0]
print(f"  pool dry (first wave {us(d.min()):.1f}, median {np.median(us(d)):.1f}, last {us(d.max()):.1f}) us")
print(f"  wave end: first {us(t1.min()):.1f}, 10% {np.percentile(us(t1), 10):.1f}, median {np.median(us(t1)):.1f}, 90% {np.percentile(us(t1), 90):.1f}, 99% {np.percentile(us(t1), 99):.1f}, last {us(t1.max()):.1f} us")
cam = (raw[:, 3] & 0xFFFFFFFF).astype(np.int64); bnc = (raw[:, 3] >> 32).astype(np.int64); spl = (raw[:, 4] & 0xFFFFFFFF).astype(np.int64); lr = (raw[:, 4] >> 32).astype(np.int64)
print(f"  per wave: camera rounds {cam.mean():.1f} (min {cam.min()} max {cam.max()}), bounce rounds {bnc.mean():.1f} (min {bnc.min()} max {bnc.max()}), split rounds {spl.mean():.2f}, lanes per round {lr.sum() / max(1, (cam + bnc).sum()):.1f}")
print(f"  triangles walked per bounce round (the union of the lanes' rows): {raw[:, 5].sum() / max(1, bnc.sum()):.1f} of {tris.shape[0]}")
busy = (t1 - t0).astype(np.float64) / 100.0
print(f"  wave busy time: mean {busy.mean():.1f} us, min {busy.min():.1f}, max {busy.max():.1f}; idle share of the span {1 - busy.mean() / us(t1.max()):.3f}")
if phases.sum() > 0:
    names = ["loop head + claims", "camera round set-up (park, decode, begin_sample)", "bounce round set-up (unpark)", "camera walk (rectangles + tests)", "bounce culls (row union, leaf boxes)",
             "bounce triangle tests", "shade + sample store", "other walks"]
    tot = phases.sum()
    print("  wave time by phase (shader clocks, all waves): " + "; ".join(f"{n} {phases[:, k].sum() / tot:.3f}" for k, n in enumerate(names) if phases[:, k].sum() > 0))
    print(f"  clocks per wave {phases.sum(axis=1).mean():.0f} = {phases.sum(axis=1).mean() / max(busy.mean(), 1e-9) / 1e3:.2f} GHz x busy time; per camera round: set-up {phases[:, 1].sum() / max(1, cam.sum()):.0f}, walk {phases[:, 3].sum() / max(1, cam.sum()):.0f}; "
          f"per bounce round: set-up {phases[:, 2].sum() / max(1, bnc.sum()):.0f}, culls {phases[:, 4].sum() / max(1, bnc.sum()):.0f}, tests {phases[:, 5].sum() / max(1, bnc.sum()):.0f}; shade + store per round {phases[:, 6].sum() / max(1, (cam + bnc).sum()):.0f}; loop head per round {phases[:, 0].sum() / max(1, (cam + bnc).sum()):.0f}")
hist, edges = np.histogram(us(t1), bins=12)
print("  end-time histogram (us):", " ".join(f"{edges[i]:.0f}:{hist[i]}" for i in range(len(hist))))

after = (raw[:, 6] & 0xFFFFFFFF).astype(np.int64); lanes_after = (raw[:, 6] >> 32).astype(np.int64); last_cam = raw[:, 7].astype(np.int64)
tail = (t1 - np.where(dry > 0, dry, t1)).astype(np.float64) / 100.0
print(f"  after the pool is dry: rounds per wave {after.mean():.2f} (max {after.max()}), lanes per such round {lanes_after.sum() / max(1, after.sum()):.1f}; time from dry to exit: median {np.median(tail):.1f} us, 90% {np.percentile(tail, 90):.1f}, max {tail.max():.1f}")
lc = np.where(last_cam > 0, us(last_cam), 0.0)
print(f"  last camera round of a wave: median {np.median(lc):.1f} us, 90% {np.percentile(lc, 90):.1f}, 99% {np.percentile(lc, 99):.1f}, last {lc.max():.1f} us; exit - last camera round: median {np.median(us(t1) - lc):.1f}, 90% {np.percentile(us(t1) - lc, 90):.1f}, max {(us(t1) - lc).max():.1f} us")
order = np.argsort(us(t1))[-8:]
for w in order:
    print(f"    late wave {w} (block {w // 4}): start {us(t0[w]):.1f} last camera round {lc[w]:.1f} dry {us(dry[w]) if dry[w] > 0 else -1:.1f} end {us(t1[w]):.1f} us; rounds camera {cam[w]} bounce {bnc[w]} split {spl[w]}, after dry {after[w]}")
