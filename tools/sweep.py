#!/usr/bin/env python3
"""One parametrised sweep driver for the GPU box (replaces the per-question sweep_*.sh / ab_*.sh scripts of rounds 1-4, kept under tools/archive/ with the
command line that reproduces each).  A sweep is a GRID of environment knobs and / or build variants, times a list of bench.py workloads:

  tools/sweep.py --env RVPT_HIP_BLOCKS_PER_CU=2,3,5 --env RVPT_HIP_CLAIM_UNITS=8,32 \\
                 --workload "k20:--steps 20 --warmup 5" --workload "share3:--steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3" [--reps 2]
  tools/sweep.py --variant base: --variant pad9:-DRV_BVH4_TOP_QUADS=9 --workload "c3:--scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16"

--env NAME=v1,v2,...   one axis of the grid (the value `-` leaves the variable unset)
--variant tag:FLAGS    build librvpt_hip.so with extra hipcc FLAGS into build/exp/<tag>.so (built where the script runs; hipcc is on the GPU box) and run
                       every grid point against it (RVPT_HIP_LIB)
--workload tag:ARGS    bench.py arguments (--no-cpu-baseline is added); named workloads: k20 k200 c3 c4 c5 default_bvh share<r>of<n>
--reps N               repeat the whole grid N times (back to back: A/B on one box)
--out FILE             also append the table to FILE (e.g. gpurun_out/<name>.txt -> profiles/)
Prints one line per (variant, grid point): every workload's Msamples/s (or ms per frame for emulated shares) and ms per step."""
import argparse
import itertools
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
NAMED = {
    "k20": "--steps 20 --warmup 5", "k200": "--steps 200 --warmup 20",
    "c3": "--scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16", "c4": "--scene heightfield --traversal bvh --steps 96 --warmup 16",
    "c5": "--scene cornell --width 3840 --height 2160 --aa 16 --traversal bvh --steps 16 --warmup 4 --batch 4",
    "default_bvh": "--traversal bvh --steps 96 --warmup 16",
}


def workload_args(spec):
    tag, _, rest = spec.partition(":")
    if rest:
        return tag, rest.split()
    if tag in NAMED:
        return tag, NAMED[tag].split()
    if tag.startswith("share") and "of" in tag:  # share3of8[:base workload]
        r, n = tag[5:].split("of")
        return tag, NAMED["k20"].split() + ["--emulate-world", n, "--emulate-rank", r]
    raise SystemExit(f"unknown workload {spec!r}")


def build_variant(tag, flags):
    from rvpt_amd import build as B
    out = ROOT / "build" / "exp" / f"{tag}.so"
    out.parent.mkdir(parents=True, exist_ok=True)
    if not out.exists() or out.stat().st_mtime < max(p.stat().st_mtime for p in B.SOURCES + B.HEADERS):
        # the laboratory build (include/rvpt_hip_lab.h) without its internal checks: the knobs a sweep turns exist only there
        subprocess.run([B.hipcc(), *B.FLAGS, "-DRVPT_HIP_LAB=1", *flags, *map(str, B.LAB_SOURCES), "-o", str(out)], check=True)
    return out


def run_bench(args, env):
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", *args], env=env, capture_output=True, text=True)
    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if not line:
        return "FAILED"
    d = json.loads(line[-1])
    return f"{d.get('value', d.get('ms_per_frame_wall'))} ({d.get('ms_per_step', d.get('kernel_ms'))})"


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--workload", action="append", default=[])
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    axes = [(e.split("=", 1)[0], e.split("=", 1)[1].split(",")) for e in a.env]
    workloads = [workload_args(w) for w in (a.workload or ["k20"])]
    variants = [(v.partition(":")[0], v.partition(":")[2].split()) for v in a.variant] or ([("lab", [])] if axes else [("", None)])
    libs = [(tag, build_variant(tag, flags) if flags is not None else None) for tag, flags in variants]
    lines = []
    for rep in range(a.reps):
        for tag, lib in libs:
            for point in itertools.product(*[vals for _, vals in axes]) if axes else [()]:
                env = dict(os.environ)
                if lib is not None:
                    env["RVPT_HIP_LIB"] = str(lib)
                label = [tag] if tag else []
                for (name, _), val in zip(axes, point):
                    if val == "-":
                        env.pop(name, None)
                    else:
                        env[name] = val
                    label.append(f"{name.replace('RVPT_HIP_', '').lower()}={val}")
                cells = [f"{wtag} {run_bench(wargs, env)}" for wtag, wargs in workloads]
                line = f"{' '.join(label) or 'default'}: " + " | ".join(cells)
                print(line, flush=True)
                lines.append(line)
    if a.out:
        with open(a.out, "a") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
