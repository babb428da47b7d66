#!/bin/bash
# Run on the GPU box: knobs of the wavefront traverse kernel on C3 (Cornell 1080p x 4 spp).  -> gpurun_out/wf_knobs.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/wf_knobs.txt
: > $OUT
one() {  # label, env assignments...
  local label=$1; shift
  line=$(env "$@" timeout 300 python $REPO/bench.py --no-cpu-baseline --scene ${SCENE:-cornell} --traversal bvh --aa ${AA:-4} --steps ${STEPS:-16} --warmup 8 --wavefront on 2>&1 | tail -1)
  echo "$label $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'grid', d['config']['grid_blocks'], 'lds', d['config']['lds_bytes_per_block'])" 2>/dev/null || echo FAILED)" | tee -a $OUT
}
one base X=1
for r in 4 8 32 48 64; do one refill$r RVPT_HIP_BVH_REFILL=$r; done
for l in 3 4 6 8 12; do one stack$l RVPT_HIP_BVH_STACK_LDS=$l; done
for b in 2 3 4 6; do one bpc$b RVPT_HIP_BLOCKS_PER_CU=$b; done
for t in 0 64 128 512; do one top$t RVPT_HIP_BVH_TOP_NODES=$t; done
for lb in 1 8 32; do one leafbatch$lb RVPT_HIP_BVH_LEAF_BATCH=$lb; done
one inflight1 RVPT_HIP_FRAMES_IN_FLIGHT=1
one inflight2 RVPT_HIP_FRAMES_IN_FLIGHT=2
