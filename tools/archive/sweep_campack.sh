#!/bin/bash
# (historic: the binary-tree camera-packet instances this script measures were retired — apply profiles/r04_exp_campack_binary.patch to reproduce; results: profiles/r04_campack*.txt, r04_ab_wide_resident.txt)
# Run on the GPU box: camera packets (trace_bvh<..., CAMPACK>) against the per-lane walk on the three BVH workloads, then the two knobs
# (lanes needed to form a packet, RVPT_HIP_BVH_CAM_MIN; lanes at or below which a node's lanes leave the packet, RVPT_HIP_BVH_DETACH) and the
# refill threshold.  One box, A/B back to back.  -> gpurun_out/campack.txt
# usage: tools/archive/sweep_campack.sh [quick]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/campack.txt
mkdir -p $REPO/gpurun_out; : > $OUT
val() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['segments_per_sample'])" 2>/dev/null || echo FAILED; }
one() {  # label, env..., -- bench args
  label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  line=$(env "${envs[@]}" timeout 600 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 "$@" 2>/dev/null | tail -1 | val)
  echo "$label ${envs[*]} : $line" | tee -a $OUT
}
DEF="--traversal bvh --steps 296 --warmup 32"
C3="--scene cornell --aa 4 --traversal bvh --steps 64 --warmup 16"
C4="--scene heightfield --traversal bvh --steps 160 --warmup 32"
for rep in 1 2; do
  one default_perlane X=1 -- $DEF --per-lane;  one default_campack X=1 -- $DEF
  one c3_perlane X=1 -- $C3 --per-lane;        one c3_campack X=1 -- $C3
  one c4_perlane X=1 -- $C4 --per-lane;        one c4_campack X=1 -- $C4
done
[ "$1" = quick ] && exit 0
for det in 0 2 4 8 12 16 24 32 48; do
  one default RVPT_HIP_BVH_DETACH=$det -- $DEF
  one c3 RVPT_HIP_BVH_DETACH=$det -- $C3
  one c4 RVPT_HIP_BVH_DETACH=$det -- $C4
done
for cm in 8 16 24 32 48 64; do
  one default RVPT_HIP_BVH_CAM_MIN=$cm -- $DEF
  one c3 RVPT_HIP_BVH_CAM_MIN=$cm -- $C3
  one c4 RVPT_HIP_BVH_CAM_MIN=$cm -- $C4
done
for rf in 16 32 48 64; do
  one c3 RVPT_HIP_BVH_REFILL=$rf -- $C3
  one c4 RVPT_HIP_BVH_REFILL=$rf -- $C4
  one c3 RVPT_HIP_BVH_REFILL=$rf RVPT_HIP_BVH_CAM_MIN=$rf -- $C3
  one c4 RVPT_HIP_BVH_REFILL=$rf RVPT_HIP_BVH_CAM_MIN=$rf -- $C4
done
