#!/bin/bash
# Run on the GPU box: work-groups per CU x launches in flight for the wide-tree BVH kernel (C3, C4 geometry).  -> gpurun_out/wide_flight.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wide_flight.txt
mkdir -p $REPO/gpurun_out; : > $OUT
one() {  # label, env assignments...
  local label=$1; shift
  a=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --scene cornell --traversal bvh --aa 4 --steps 48 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  b=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --scene heightfield --traversal bvh --aa 1 --steps 96 --warmup 16 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED)
  echo "$label c3 $a c4geo $b" | tee -a $OUT
}
one base X=1
for f in 3 4 6; do for b in 2 3 4 5 6; do one "inflight$f bpc$b" RVPT_HIP_FRAMES_IN_FLIGHT=$f RVPT_HIP_BLOCKS_PER_CU=$b; done; done
one base X=1
for bt in 4 16; do one "batch$bt" X=1; done
