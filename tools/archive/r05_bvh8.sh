#!/bin/bash
# round 5: the 8-wide BVH walk (rvpt_bvh8.hip, RVPT_HIP_BVH_WIDE8=1) against the 4-wide default — parity first, then C3 / C4 geometry A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
RVPT_HIP_BVH_WIDE8=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "bvh or wide or million or reference or heightfield or cornell or fixtures or sweep or stack or leaf" > gpurun_out/r05_bvh8_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r05_bvh8_tests.log | tail -3
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['lds_bytes_per_block'], d['config']['grid_blocks'])"; }
for rep in 1 2; do
for e in "RVPT_HIP_BVH_WIDE8=0" "RVPT_HIP_BVH_WIDE8=1" "RVPT_HIP_BVH_WIDE8=1 RVPT_HIP_BVH_TOP_NODES=16" "RVPT_HIP_BVH_WIDE8=1 RVPT_HIP_BVH_TOP_NODES=64" "RVPT_HIP_BVH_WIDE8=1 RVPT_HIP_BVH_STACK_LDS=12"; do
  echo "$e: C3 $(env $e bash -c "$(declare -f one); one --scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16") | C4 $(env $e bash -c "$(declare -f one); one --scene heightfield --traversal bvh --steps 96 --warmup 16")"
done
done
