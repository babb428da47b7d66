#!/usr/bin/env python3
"""GPU-box sweep of the work-distribution knobs (env overrides read by rvpt_abi.hip) for N = 1 and emulated
N-way shares.  Prints ms per frame (wall, steady pipeline)."""
import itertools, json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
worlds = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
combos = []
grid = os.environ.get("SWEEP", "2,3,4;2,4,8;2,4,8;2,3")
axes = [[int(v) for v in a.split(",")] for a in grid.split(";")]
for bpc, first, claim, depth in itertools.product(*axes):
    combos.append(dict(RVPT_HIP_BLOCKS_PER_CU=bpc, RVPT_HIP_FIRST_UNITS=first, RVPT_HIP_CLAIM_UNITS=claim, RVPT_HIP_FRAMES_IN_FLIGHT=depth))
print("bpc first claim depth | " + " ".join(f"N={w:<6d}" for w in worlds))
for c in combos:
    env = dict(os.environ, **{k: str(v) for k, v in c.items()})
    row = []
    for w in worlds:
        args = [sys.executable, str(ROOT / "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline"]
        if w > 1:
            args += ["--emulate-world", str(w)]
        out = subprocess.run(args, env=env, capture_output=True, text=True).stdout.strip().splitlines()
        try:
            j = json.loads(out[-1])
            row.append(j["ms_per_step"] if w == 1 else j["rank0_ms_per_frame_wall"])
        except Exception:
            row.append(float("nan"))
    print(f"{c['RVPT_HIP_BLOCKS_PER_CU']:3d} {c['RVPT_HIP_FIRST_UNITS']:5d} {c['RVPT_HIP_CLAIM_UNITS']:5d} {c['RVPT_HIP_FRAMES_IN_FLIGHT']:5d} | " + " ".join(f"{v:8.4f}" for v in row), flush=True)
