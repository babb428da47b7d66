#!/bin/bash
# GPU box: every BASELINE-style configuration on one GPU -> gpurun_out/config_table.jsonl (one bench.py line each).
# usage: tools/bench_table.sh [extra bench.py args]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/config_table.jsonl
: > $OUT
run() { echo "## $*" >&2; python $REPO/bench.py "$@" 2>/dev/null | tail -1 >> $OUT; }
CPU="--cpu-seconds 6"
run --width 256 --height 256 --steps 400 --warmup 40 $CPU
run --width 256 --height 256 --steps 400 --warmup 40 --traversal bvh $CPU
run --steps 20 --warmup 5 $CPU
run --steps 296 --warmup 32 $CPU
run --steps 296 --warmup 32 --batch 1 --no-cpu-baseline
run --steps 296 --warmup 32 --traversal bvh --batch 1 --no-cpu-baseline
run --steps 296 --warmup 32 --traversal bvh $CPU
run --steps 296 --warmup 32 --traversal bvh_ordered --no-cpu-baseline
run --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16 --batch 1 --no-cpu-baseline
run --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16 $CPU
run --scene cornell --aa 4 --traversal bvh_ordered --steps 96 --warmup 16 --no-cpu-baseline
run --scene cornell --aa 4 --traversal brute --steps 4 --warmup 1 --no-cpu-baseline
run --scene heightfield --traversal bvh --steps 296 --warmup 32 --batch 1 --no-cpu-baseline
run --scene heightfield --traversal bvh --steps 296 --warmup 32 $CPU
run --scene heightfield --traversal bvh_ordered --steps 296 --warmup 32 --no-cpu-baseline
run --scene cornell --width 3840 --height 2160 --aa 16 --traversal bvh --steps 16 --warmup 4 --batch 4 --no-cpu-baseline
# round 4: the binary per-lane walk (rounds 1-3's kernel) beside the defaults above (camera packets / the wide tree)
run --steps 296 --warmup 32 --traversal bvh --per-lane --no-cpu-baseline
run --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16 --per-lane --no-cpu-baseline
run --scene heightfield --traversal bvh --steps 296 --warmup 32 --per-lane --no-cpu-baseline
run --scene cornell --width 3840 --height 2160 --aa 16 --traversal bvh --steps 16 --warmup 4 --batch 4 --per-lane --no-cpu-baseline
run --scene cornell --width 3840 --height 2160 --aa 16 --traversal bvh_ordered --steps 16 --warmup 4 --batch 4 --no-cpu-baseline
python - <<PY
import json
for l in open("$OUT"):
    j = json.loads(l)
    c = j["config"]
    print(f'{c["workload"][:95]:95s} b{c["frames_per_dispatch"]:<2d} {j["value"]:10.1f} Msamples/s {j["ms_per_step"]:9.4f} ms  seg/sample {c["segments_per_sample"]:.3f}  cpu {j.get("cpu_baseline", {}).get("value")}')
PY
