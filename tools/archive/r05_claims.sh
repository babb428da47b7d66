#!/bin/bash
# round 5: the work-claim bottleneck of the packet kernel after the culls — shards x claim size, full frame and an 8-way share
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value', d.get('ms_per_frame_wall')), d.get('ms_per_step', d.get('kernel_ms')))"; }
for lib in base shards16 shards32 shards64; do
  for cu in 8 16 32 64; do
    export RVPT_HIP_LIB=$PWD/build/exp/$lib.so RVPT_HIP_CLAIM_UNITS=$cu
    echo "$lib claim $cu: k20 $(one --steps 20 --warmup 5) | k200 $(one --steps 200 --warmup 20) | share 3/8 k20 $(one --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3)"
  done
done
