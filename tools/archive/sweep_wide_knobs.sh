#!/bin/bash
# Run on the GPU box: one-at-a-time knob sweep of the wide-tree BVH kernel (rvpt_bvh4.hip) on C3 (Cornell 1080p x 4 spp) and C4 geometry (1 M-triangle
# terrain 1080p x 1 spp), and library builds under build/exp (RV_BVH4_MIN_WAVES=6: 80 VGPRs + one spilled register for a sixth wave per SIMD).
# -> gpurun_out/wide_knobs.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wide_knobs.txt
mkdir -p $REPO/gpurun_out; : > $OUT
one() {  # label, env assignments...
  local label=$1; shift
  a=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --scene cornell --traversal bvh --aa 4 --steps 32 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  b=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --scene heightfield --traversal bvh --aa 1 --steps 96 --warmup 16 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  echo "$label c3 $a c4geo $b" | tee -a $OUT
}
one base X=1
one w6 RVPT_HIP_LIB=$REPO/build/exp/librvpt_w6.so
one base X=1
one w6 RVPT_HIP_LIB=$REPO/build/exp/librvpt_w6.so
for r in 16 24 40 48; do one refill$r RVPT_HIP_BVH_REFILL=$r; done
for lb in 8 12 24 32; do one leafbatch$lb RVPT_HIP_BVH_LEAF_BATCH=$lb; done
for b in 2 4 5; do one bpc$b RVPT_HIP_BLOCKS_PER_CU=$b; done
for f in 2 4 6; do one inflight$f RVPT_HIP_FRAMES_IN_FLIGHT=$f; done
for t in 0 16 32 128 192; do one top$t RVPT_HIP_BVH_TOP_NODES=$t; done
for l in 4 6 10 12; do one stack$l RVPT_HIP_BVH_STACK_LDS=$l; done
one top128_stack6 RVPT_HIP_BVH_TOP_NODES=128 RVPT_HIP_BVH_STACK_LDS=6
one top128_stack4 RVPT_HIP_BVH_TOP_NODES=128 RVPT_HIP_BVH_STACK_LDS=4
one base X=1
# the default scene (LDS-resident instances by default) through the wide kernel instead: nodes of the wide tree in LDS, triangles from L2
dflt() { local label=$1; shift; echo "$label default-scene bvh $(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --traversal bvh --steps 296 --warmup 32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block'])" 2>/dev/null || echo FAILED)" | tee -a $OUT; }
dflt resident X=1
dflt wide_top64 RVPT_HIP_BVH_NO_RESIDENT=1
dflt wide_top128 RVPT_HIP_BVH_NO_RESIDENT=1 RVPT_HIP_BVH_TOP_NODES=128
dflt wide_top128_bpc5 RVPT_HIP_BVH_NO_RESIDENT=1 RVPT_HIP_BVH_TOP_NODES=128 RVPT_HIP_BLOCKS_PER_CU=5
dflt wide_top128_stack4 RVPT_HIP_BVH_NO_RESIDENT=1 RVPT_HIP_BVH_TOP_NODES=128 RVPT_HIP_BVH_STACK_LDS=4
dflt resident X=1
