#!/bin/bash
# Run on the GPU box: one-at-a-time knob sweep of the HBM-resident BVH megakernel on C3 (Cornell 1080p x 4 spp) and C4 geometry
# (1 M-triangle terrain 1080p x 1 spp), bench.py's default 8 frames per launch.   -> gpurun_out/mega_knobs.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mega_knobs.txt
: > $OUT
one() {  # label, env assignments...
  local label=$1; shift
  a=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --steps 24 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  b=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --scene heightfield --traversal bvh --aa 1 --steps 64 --warmup 16 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  echo "$label c3 $a c4geo $b" | tee -a $OUT
}
one base X=1
one base X=1
for r in 16 24 40 48; do one refill$r RVPT_HIP_BVH_REFILL=$r; done
for lb in 8 12 24 32; do one leafbatch$lb RVPT_HIP_BVH_LEAF_BATCH=$lb; done
for b in 2 4; do one bpc$b RVPT_HIP_BLOCKS_PER_CU=$b; done
for f in 2 4 6; do one inflight$f RVPT_HIP_FRAMES_IN_FLIGHT=$f; done
for t in 128 512 1024; do one top$t RVPT_HIP_BVH_TOP_NODES=$t; done
for l in 6 10; do one stack$l RVPT_HIP_BVH_STACK_LDS=$l; done
one base X=1
