#!/bin/bash
# Run on the GPU box: launches in flight x work-groups per CU for the HBM-resident BVH megakernel (C3, C4 geometry, C3 at one frame per launch).  -> gpurun_out/mega_flight.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/mega_flight.txt
: > $OUT
val() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])" 2>/dev/null || echo FAILED; }
for f in 3 4 6 8; do for b in 1 2 3; do
  a=$(RVPT_HIP_FRAMES_IN_FLIGHT=$f RVPT_HIP_BLOCKS_PER_CU=$b timeout 300 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --steps 48 --warmup 16 2>/dev/null | val)
  c=$(RVPT_HIP_FRAMES_IN_FLIGHT=$f RVPT_HIP_BLOCKS_PER_CU=$b timeout 300 python $REPO/bench.py --no-cpu-baseline --scene heightfield --traversal bvh --aa 1 --steps 96 --warmup 16 2>/dev/null | val)
  d=$(RVPT_HIP_FRAMES_IN_FLIGHT=$f RVPT_HIP_BLOCKS_PER_CU=$b timeout 300 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --batch 1 --steps 24 --warmup 8 2>/dev/null | val)
  echo "inflight $f bpc $b : c3 $a  c4geo $c  c3_batch1 $d" | tee -a $OUT
done; done
a=$(timeout 300 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --steps 48 --warmup 16 2>/dev/null | val)
c=$(timeout 300 python $REPO/bench.py --no-cpu-baseline --scene heightfield --traversal bvh --aa 1 --steps 96 --warmup 16 2>/dev/null | val)
d=$(timeout 300 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --batch 1 --steps 24 --warmup 8 2>/dev/null | val)
echo "default policy : c3 $a  c4geo $c  c3_batch1 $d" | tee -a $OUT
