#!/bin/bash
# Run on the GPU box: BVH configurations (C3 / C4 geometry / C5 shape), megakernel against the wavefront pipeline.  -> gpurun_out/bench_wf.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/bench_wf.txt
: > $OUT
run() {  # label, args...
  local label=$1; shift
  line=$(timeout 600 python $REPO/bench.py --no-cpu-baseline "$@" 2>&1 | tail -1)
  echo "$label $line" | python -c "
import sys, json
l = sys.stdin.read().strip(); label, _, js = l.partition(' ')
try:
    d = json.loads(js); print(label, d['value'], 'Msamples/s', 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'variant-grid', d['config']['grid_blocks'], 'S', d['config']['segments_per_sample'], 'B', d['config']['frames_per_dispatch'], 'hbm_GBs', d['roofline'].get('achieved'))
except Exception as e:
    print(label, 'FAILED', js[-300:])
" | tee -a $OUT
}
for wf in ${WF_MODES:-off on}; do
  run c3_$wf --scene cornell --traversal bvh --aa 4 --steps 32 --warmup 8 --wavefront $wf
  run c4geo_$wf --scene heightfield --traversal bvh --aa 1 --steps 64 --warmup 16 --wavefront $wf
  run c3ord_$wf --scene cornell --traversal bvh_ordered --aa 4 --steps 32 --warmup 8 --wavefront $wf
done
