#!/usr/bin/env python3
"""Stress check (GPU box): thousands of frames dispatched back to back with frames in flight must produce exactly the
accumulator of the serial (one frame in flight, fused blend) run — catches ordering / counter-reset races that the
short parity tests could miss; the same frames issued in batches (rvpt_hip_dispatch_frames) must give it too.
python tools/soak.py [frames] [width] [height]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    from rvpt_amd import RenderSettings, native, scene
    frames, W, H, trav, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    tris, mats = scene.materials_showcase_scene()
    nodes, idx = native.build_bvh(tris)
    cam = np.zeros(20, np.float32); cam[[0, 5, 10, 15]] = 1; cam[13] = 1.0; cam[14] = -2.5; cam[16] = W / H; cam[17] = 1.5707964; cam[18] = 4
    ctx = native.Context(W, H, 0, 0, 1, native.COUNT_SEGMENTS | (native.TRAVERSAL_BVH if trav == "bvh" else 0))
    ctx.upload_scene(nodes if trav == "bvh" else None, tris[idx], mats)
    batched = os.environ.get("SOAK_BATCHED") == "1"
    f = 0
    while f < frames:  # aa changes every 5 frames; batches (rvpt_hip_dispatch_frames) never straddle a change
        n = min(1 + (f * 7) % 4, 5 - f % 5, frames - f) if batched else 1
        ctx.set_frame(RenderSettings(aa=1 + ((f // 5) % 3), current_frame=f).pack(), cam)
        ctx.dispatch() if n == 1 else ctx.dispatch_frames(n)
        if f % 97 == 0:
            ctx.query()
        f += n
    img = ctx.read()
    np.save(out, img)
    print(trav, os.environ.get("RVPT_HIP_FRAMES_IN_FLIGHT", "default"), ctx.stats(), float(img.mean()))
    ctx.close()
    sys.exit(0)

import numpy as np
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 320
H = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ok = True
for trav in ("brute", "bvh"):
    outs = []
    for depth, batched in (("1", "0"), ("3", "0"), ("6", "0"), ("3", "1"), ("1", "1")):
        out = f"/tmp/soak_{trav}_{depth}_{batched}.npy"
        env = dict(os.environ, RVPT_HIP_FRAMES_IN_FLIGHT=depth, SOAK_BATCHED=batched)
        subprocess.run([sys.executable, __file__, "--child", str(frames), str(W), str(H), trav, out], env=env, check=True)
        outs.append(np.load(out))
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    print(trav, "frames in flight 1 / 3 / 6, frame by frame and in batches: identical:", same, "finite:", bool(np.isfinite(outs[0]).all()))
    ok &= same
sys.exit(0 if ok else 1)
