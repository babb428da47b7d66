#!/bin/bash
# Run on the GPU box: streamed brute-force kernel, windows per wave (RV_STREAM_DEPTH builds under build/exp/).  -> gpurun_out/stream_depth.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/stream_depth.txt
: > $OUT
cd $REPO
one() {  # label, lib, bench args...
  local label=$1 lib=$2; shift 2
  line=$(RVPT_HIP_LIB=$lib timeout 900 python bench.py --no-cpu-baseline --traversal brute "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s', d['roofline'].get('ray_triangle_tests_per_s'), 'tests/s', 'kernel_ms', d['roofline']['hbm']['kernel_ms'])" 2>/dev/null || echo FAILED)" | tee -a $OUT
}
for rep in 1 2; do
for d in 2 3 4; do
  one "rep$rep depth$d terrain1M_512x288" $REPO/build/exp/librvpt_depth$d.so --scene heightfield --width 512 --height 288 --steps 4 --warmup 2 --batch 2 --ramp-seconds 0
  one "rep$rep depth$d cornell9k_1080p_aa4" $REPO/build/exp/librvpt_depth$d.so --scene cornell --aa 4 --steps 4 --warmup 2 --batch 2 --ramp-seconds 0
done
done
# the regime where the stream does not fit L2 at full occupancy: 1 M triangles at 1920x1080 (2.5 s per frame)
if [ -n "${FULL_HD:-}" ]; then
for d in 2 3 4; do
  one "depth$d terrain1M_1080p" $REPO/build/exp/librvpt_depth$d.so --scene heightfield --steps 2 --warmup 1 --batch 1 --ramp-seconds 0
done
fi
