#!/usr/bin/env python3
"""Build experimental variants of librvpt_hip.so (extra -D / compiler flags) into build/exp/ and, on a GPU
box, bench each one:   python tools/archive/exp_variants.py build|bench [bench.py args...]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from rvpt_amd import build as B  # noqa: E402

VARIANTS = {
    "base": [],
    "shards16": ["-DRV_CLAIM_SHARDS=16"],
    "shards32": ["-DRV_CLAIM_SHARDS=32"],
    "shards64": ["-DRV_CLAIM_SHARDS=64"],
} if os.environ.get("EXP_SET") == "shards" else {
    "base": [],
    "sched_ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "sched_mem": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "no_postsched": ["-mllvm", "-enable-post-misched=0"],
    "O2": ["-O2"],
}
OUT = ROOT / "build" / "exp"


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    for name, extra in VARIANTS.items():
        cmd = [B.hipcc(), *B.FLAGS, *extra, *map(str, B.SOURCES), "-o", str(OUT / f"{name}.so")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        print("built" if r.returncode == 0 else "FAILED " + r.stderr[-300:], name)


def bench(argv):
    for name in VARIANTS:
        env = dict(os.environ, RVPT_HIP_LIB=str(OUT / f"{name}.so"))
        res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", *argv], env=env,
                             capture_output=True, text=True)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", res.stderr[-400:])
            continue
        j = json.loads(line[-1])
        print(f"{name:12s} {j['value']:9.1f} Msamples/s  ms/step {j['ms_per_step']:.5f}  grid {j['config']['grid_blocks']}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        bench(sys.argv[2:])
