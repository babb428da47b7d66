#!/usr/bin/env python3
"""GPU-box sweep: frames in flight x work-groups per CU (env overrides read by rvpt_abi.hip).  usage: sweep_flight.py scene traversal"""
import itertools, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
scene, trav = sys.argv[1], sys.argv[2]
grid = os.environ.get("SWEEP", "3,4,6,8;1,2,3,4")
axes = [[int(v) for v in a.split(",")] for a in grid.split(";")]
print("flight bpc | ms/frame", flush=True)
for fl, bpc in itertools.product(*axes):
    env = dict(os.environ, RVPT_HIP_BLOCKS_PER_CU=str(bpc), RVPT_HIP_FRAMES_IN_FLIGHT=str(fl))
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "96", "--warmup", "16", "--no-cpu-baseline", "--scene", scene,
                          "--traversal", trav], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    try:
        ms = json.loads(out[-1])["ms_per_step"]
    except Exception:
        ms = float("nan")
    print(f"{fl:5d} {bpc:4d} | {ms:8.4f}", flush=True)
