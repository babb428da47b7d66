#!/usr/bin/env python3
"""Run on the GPU box: the packet kernel's interleaved claim order (RVPT_HIP_PACKETS_INTERLEAVE = blocks per group; 0 = tile-linear order) — the same bits whatever the
order, then the driver's command, 200 steps, one frame per launch and rank 3's share of eight for every group size.  -> stdout
usage: tools/interleave_ab.py [check|bench|all] [groups, default "0 1 2 4 8"]"""
import hashlib, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
what = sys.argv[1] if len(sys.argv) > 1 else "all"
groups = (sys.argv[2] if len(sys.argv) > 2 else "0 1 2 4 8").split()

def check():
    import numpy as np
    from rvpt_amd import RVPT, native, scene
    tris, mats = scene.default_scene()
    ref = {}
    for g in groups:
        os.environ["RVPT_HIP_PACKETS_INTERLEAVE"] = g
        for (w, h, aa, world, rank) in ((1920, 1080, 1, 1, 0), (1920, 1080, 2, 1, 0), (1000, 700, 1, 1, 0), (1920, 1080, 1, 8, 3), (640, 360, 1, 3, 1)):
            r = RVPT(w, h, device=0, traversal="brute", tile_rank=rank, tile_world=world, flags=native.COUNT_SEGMENTS)
            r.add_triangles(tris)
            for m in mats:
                r.add_material(m)
            r.render_settings.aa = aa
            r.initialize()
            r.update(); r.draw_frames(5); r.update(); r.draw(); r.update(); r.draw_frames(3); r.wait()
            info = r.context.cull_info() if hasattr(r.context, "cull_info") else -1
            img = np.ascontiguousarray(r.read_frame()) if world == 1 else None
            seg = r.context.stats()
            r.shutdown()
            key = (w, h, aa, world, rank)
            dig = hashlib.sha256(img.tobytes()).hexdigest()[:16] if img is not None else "-"
            print(f"interleave {g} {key}: cull info {info} image {dig} stats {seg}")
            if world == 1:
                if key in ref:
                    assert ref[key] == (dig, seg), f"order changes the image: {key} {g}"
                else:
                    ref[key] = (dig, seg)
    print("check: the same bits and segment counts for every claim order")

def bench():
    py = sys.executable
    def line(args, env):
        out = subprocess.run([py, str(ROOT / "bench.py"), *args], env={**os.environ, **env}, capture_output=True, text=True).stdout.strip().splitlines()
        return json.loads(out[-1]) if out else {}
    for g in groups:
        env = {"RVPT_HIP_PACKETS_INTERLEAVE": g}
        k20 = [line(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-one-frame-leg"], env).get("value") for _ in range(3)]
        d = line(["--steps", "200", "--warmup", "20", "--no-cpu-baseline"], env)
        sh = [line(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-one-frame-leg", "--emulate-world", "8", "--emulate-rank", r], env) for r in ("3", "5")]
        one = subprocess.run([py, str(ROOT / "tools" / "one_frame_per_launch.py")], env={**os.environ, **env}, capture_output=True, text=True).stdout.strip()
        print(f"interleave {g}: k20 {k20} k200 {d.get('value')} one-frame leg {d.get('value_one_frame_per_launch', {}).get('value')} share-of-8 ms/frame {[s.get('ms_per_frame_wall') for s in sh]} kernel ms {[s.get('kernel_ms') for s in sh]}")
        print(f"    {one}", flush=True)

if what in ("check", "all"):
    check()
if what in ("bench", "all"):
    bench()
