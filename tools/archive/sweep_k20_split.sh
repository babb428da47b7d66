#!/bin/bash
# Run on the GPU box: the driver's exact command (--steps 20 --warmup 5) on rank 0's share of an N-way tile partition (single-GPU
# emulation, no gather), round 2's launch rule against the split rule (renderer.launch_sizes).   -> gpurun_out/k20_split.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/k20_split.txt
: > $OUT
for rep in 1 2 3; do
for n in 1 2 4 8; do
  for split in "--no-launch-split" ""; do
    if [ $n = 1 ]; then
      line=$(python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline $split 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_frame': d['ms_per_step'], 'launches': d['config']['launches']}))")
    else
      line=$(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n $split 2>/dev/null | tail -1)
    fi
    echo "rep $rep world $n ${split:-split} $line" | tee -a $OUT
  done
done
done
