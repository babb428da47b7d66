#!/bin/bash
# round 5 (VERDICT r4 #4c): cost-ordered claims for the BVH kernels — parity, then shares of an 8-way partition and the full frame with and without
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py -x -q -k "cost_ordered or partition or dispatch_frames_equals or long_accumulation" 2>&1 | grep -E "passed|failed"
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value', d.get('ms_per_frame_wall')), d.get('ms_per_step', d.get('kernel_ms')))"; }
for co in 0 1; do
  export RVPT_HIP_COST_ORDER=$co
  echo "cost_order=$co: C4 k20 $(one --scene heightfield --traversal bvh --steps 20 --warmup 5) | C4 k96 $(one --scene heightfield --traversal bvh --steps 96 --warmup 16) | C3 k20 $(one --scene cornell --aa 4 --traversal bvh --steps 20 --warmup 5) | C3 k96 $(one --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16) | default-bvh k96 $(one --traversal bvh --steps 96 --warmup 16)"
  for r in 2 5; do
    echo "cost_order=$co: C4 share $r/8: $(one --scene heightfield --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r) $(one --scene heightfield --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r) $(one --scene heightfield --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r) | C3 share $r/8: $(one --scene cornell --aa 4 --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r) $(one --scene cornell --aa 4 --traversal bvh --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $r)"
  done
done
