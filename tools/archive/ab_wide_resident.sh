#!/bin/bash
# (historic: the binary-tree camera-packet instances this script measures were retired — apply profiles/r04_exp_campack_binary.patch to reproduce; results: profiles/r04_campack*.txt, r04_ab_wide_resident.txt)
# Run on the GPU box: the default scene (LDS-resident) over the wide tree (trace_bvh4_resident, RVPT_HIP_BVH_WIDE_RESIDENT=1) against the binary
# camera-packet kernel that is the default there.  -> gpurun_out/ab_wide_resident.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_wide_resident.txt
mkdir -p $REPO/gpurun_out; : > $OUT
one() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label ${envs[*]} : $(env "${envs[@]}" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --traversal bvh "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block'])" 2>/dev/null || echo FAILED)" | tee -a $OUT; }
L="--steps 296 --warmup 32"
for rep in 1 2; do one campack X=1 -- $L; one wide_resident RVPT_HIP_BVH_WIDE_RESIDENT=1 -- $L; done
for b in 2 3 4; do one wide_resident RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BLOCKS_PER_CU=$b -- $L; done
for st in 4 6 12; do one wide_resident RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=$st -- $L; done
for rf in 32 48; do one wide_resident RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_REFILL=$rf -- $L; done
one campack_256 X=1 -- --width 256 --height 256 --steps 400 --warmup 40
one wide_resident_256 RVPT_HIP_BVH_WIDE_RESIDENT=1 -- --width 256 --height 256 --steps 400 --warmup 40
one campack_b1 X=1 -- $L --batch 1
one wide_resident_b1 RVPT_HIP_BVH_WIDE_RESIDENT=1 -- $L --batch 1
