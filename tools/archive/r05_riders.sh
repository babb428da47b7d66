#!/bin/bash
# round 5: the two riders on the 4-wide BVH walk (LDS top nodes 144 B apart, branch-free pushes) — parity, then A/B against builds without them
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "bvh or wide or million or reference or heightfield or cornell or fixtures or sweep or stack or leaf or camera_packet" > gpurun_out/r05_riders_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r05_riders_tests.log | tail -3
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for lib in neither nopad nobf riders; do
  export RVPT_HIP_LIB=$PWD/build/exp/$lib.so
  echo "$lib: C3 $(one --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16) | C4 $(one --scene heightfield --traversal bvh --steps 96 --warmup 16) | default-bvh $(one --traversal bvh --steps 96 --warmup 16)"
done
done
