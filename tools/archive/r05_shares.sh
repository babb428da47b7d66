#!/bin/bash
# round 5 (VERDICT r4 #4a/b): the SAME rank's share of an 8-way partition ten times at the driver's command — launch jitter or ownership? — and one such launch of
# the BVH kernels split into ramp / steady / tail (instrumented build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_frame_wall'], d['kernel_ms'])"; }
for r in 2 5; do
  echo "C4 geometry, rank $r of 8, ten runs (ms per frame wall, kernel ms of the one 20-frame launch):"
  for i in 1 2 3 4 5 6 7 8 9 10; do echo -n "  $(one --scene heightfield --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r);"; done; echo
done
echo "C3, rank 5 of 8, five runs:"
for i in 1 2 3 4 5; do echo -n "  $(one --scene cornell --aa 4 --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5);"; done; echo
echo "headline, rank 5 of 8, five runs:"
for i in 1 2 3 4 5; do echo -n "  $(one --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 5);"; done; echo
WIDE=1 LONE=1 BATCH=20 RANK_OF=5/8 python tools/bvh_phase_profile.py heightfield bvh 2>&1 | tail -9
WIDE=1 LONE=1 BATCH=20 AA=4 RANK_OF=5/8 python tools/bvh_phase_profile.py cornell bvh 2>&1 | tail -9
