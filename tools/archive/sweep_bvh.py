#!/usr/bin/env python3
"""GPU-box sweep of the BVH kernel knobs (env overrides read by rvpt_abi.hip).  usage: sweep_bvh.py scene traversal
SWEEP="bpc;refill;batch" lists the values per axis."""
import itertools, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
scene, trav = sys.argv[1], sys.argv[2]
grid = os.environ.get("SWEEP", "3,6;24,32;4,8,16")
axes = [[int(v) for v in a.split(",")] for a in grid.split(";")]
print("bpc refill batch | ms/frame", flush=True)
for bpc, refill, batch in itertools.product(*axes):
    env = dict(os.environ, RVPT_HIP_BLOCKS_PER_CU=str(bpc), RVPT_HIP_BVH_REFILL=str(refill), RVPT_HIP_BVH_LEAF_BATCH=str(batch))
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "256", "--warmup", "32", "--no-cpu-baseline", "--scene", scene,
                          "--traversal", trav], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    try:
        ms = json.loads(out[-1])["ms_per_step"]
    except Exception:
        ms = float("nan")
    print(f"{bpc:3d} {refill:6d} {batch:5d} | {ms:8.4f}", flush=True)
