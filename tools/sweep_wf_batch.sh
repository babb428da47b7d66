#!/bin/bash
# Run on the GPU box: frames per launch of the wavefront pipeline (bigger launches = fewer rays in the tail of every per-iteration kernel).  -> gpurun_out/wf_batch.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wf_batch.txt
: > $OUT
for wf in on off; do
for b in 2 8 16 32; do
  steps=$((b * 4))
  line=$(timeout 600 python $REPO/bench.py --no-cpu-baseline --scene cornell --traversal bvh --aa 4 --batch $b --steps $steps --warmup $b --wavefront $wf 2>&1 | tail -1)
  echo "wavefront=$wf batch=$b $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'launches', d['config']['launches'])" 2>/dev/null || echo FAILED)" | tee -a $OUT
done
done
