#!/bin/bash
# Run on the GPU box: how the driver's 20 timed steps should go out — launch shapes at N = 1 and on rank 0's share of 2 / 4 / 8-way partitions (packet kernel).  -> gpurun_out/launch_shapes.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/launch_shapes.txt
: > $OUT
ms() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_frame_wall', d.get('ms_per_step')))" 2>/dev/null || echo FAILED; }
for rep in 1 2; do
for shape in 20 10,10 7,7,6 8,8,4 5,5,5,5 4,4,4,4,4; do
  line="rep $rep launches $shape :"
  line="$line N=1 $(python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 64 --launches $shape 2>/dev/null | ms)"
  for n in 2 4 8; do line="$line  N=$n $(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n --batch 64 --launches $shape 2>/dev/null | ms)"; done
  echo "$line" | tee -a $OUT
done
done
