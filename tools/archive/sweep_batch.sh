#!/bin/bash
# Run on the GPU box: frames per launch (--batch) of the headline configuration at K = 20 and K = 200.  -> gpurun_out/batch_sweep.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/batch_sweep.txt
: > $OUT
val() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['launches'][:3])" 2>/dev/null || echo FAILED; }
for rep in 1 2; do for b in 8 10 16 20 24 32 64; do
  echo "rep $rep batch $b: K20 $(python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 --batch $b 2>/dev/null | val)  K200 $(python $REPO/bench.py --no-cpu-baseline --batch $b 2>/dev/null | val)" | tee -a $OUT
done; done
