#!/bin/bash
# round 5: whole GPU suite (summary kept) + the bench lines that matter, one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_all_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r05_all_tests.log | tail -3
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value', d.get('ms_per_frame_wall')), d.get('ms_per_step', d.get('kernel_ms')))"; }
for rep in 1 2; do
echo "k20 $(one --steps 20 --warmup 5) | k200 $(one --steps 200 --warmup 20) | share3 $(one --steps 20 --warmup 5 --emulate-world 8 --emulate-rank 3)"
echo "C3 $(one --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16) | C4 $(one --scene heightfield --traversal bvh --steps 96 --warmup 16) | default-bvh $(one --traversal bvh --steps 96 --warmup 16) | mixed $(one --mixed-packets --steps 40 --warmup 8)"
done
