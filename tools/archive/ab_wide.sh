#!/bin/bash
# Run on the GPU box: the walk over the 4-wide tree (rvpt_bvh4.hip) against the binary per-lane walk, same box, back to back.  -> gpurun_out/ab_wide.txt
# usage: tools/archive/ab_wide.sh [extra env assignments for the wide runs, e.g. RVPT_HIP_BVH_STACK_LDS=10]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_wide.txt
mkdir -p $REPO/gpurun_out
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block'])" 2>/dev/null || echo FAILED; }
one() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label ${envs[*]} : $(env "${envs[@]}" timeout 900 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 "$@" 2>/dev/null | tail -1 | val)" | tee -a $OUT; }
C3="--scene cornell --aa 4 --traversal bvh --steps 64 --warmup 16"
C4="--scene heightfield --traversal bvh --steps 160 --warmup 32"
C5="--scene cornell --width 3840 --height 2160 --aa 16 --traversal bvh --steps 8 --warmup 4 --batch 4"
for rep in 1 2; do
  one c3_binary X=1 -- $C3 --per-lane; one c3_wide X=1 "$@" -- $C3
  one c4_binary X=1 -- $C4 --per-lane; one c4_wide X=1 "$@" -- $C4
done
one c5_binary X=1 -- $C5 --per-lane; one c5_wide X=1 "$@" -- $C5
