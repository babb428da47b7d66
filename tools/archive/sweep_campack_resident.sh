#!/bin/bash
# (historic: the binary-tree camera-packet instances this script measures were retired — apply profiles/r04_exp_campack_binary.patch to reproduce; results: profiles/r04_campack*.txt, r04_ab_wide_resident.txt)
# Run on the GPU box: the LDS-resident camera-packet kernel (default scene, BVH traversal) — lanes needed to form a packet x refill threshold, nobody
# leaving the packet (RVPT_HIP_BVH_DETACH=0), against the per-lane walk.  -> gpurun_out/campack_resident.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/campack_resident.txt
mkdir -p $REPO/gpurun_out; : > $OUT
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null || echo FAILED; }
one() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label ${envs[*]} : $(env "${envs[@]}" timeout 600 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --traversal bvh --steps 296 --warmup 32 "$@" 2>/dev/null | tail -1 | val)" | tee -a $OUT; }
for rep in 1 2; do one perlane X=1 -- --per-lane; one campack_builtin X=1 --; done
for cm in 1 8 16 24 32 48 64; do for rf in 32 48 64; do
  one campack RVPT_HIP_BVH_DETACH=0 RVPT_HIP_BVH_CAM_MIN=$cm RVPT_HIP_BVH_REFILL=$rf --
done; done
for det in 0 1 2 3; do one campack RVPT_HIP_BVH_DETACH=$det RVPT_HIP_BVH_CAM_MIN=16 --; done
one perlane_256 X=1 -- --per-lane --width 256 --height 256 --steps 400 --warmup 40
one campack_256 RVPT_HIP_BVH_DETACH=0 -- --width 256 --height 256 --steps 400 --warmup 40
