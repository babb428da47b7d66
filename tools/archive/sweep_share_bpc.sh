#!/bin/bash
# Run on the GPU box: rank 0's share of 2 / 4 / 8-way partitions at the driver's command (one 20-frame launch), work-groups per CU of the packet kernel.  -> gpurun_out/share_bpc.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/share_bpc.txt
: > $OUT
ms() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('rank0_ms_per_frame_wall', d.get('ms_per_step')))" 2>/dev/null || echo FAILED; }
for rep in 1 2; do
for n in 8 4 2; do
  line="rep $rep world $n:"
  for b in 1 2 3 4 5; do line="$line bpc$b $(RVPT_HIP_BLOCKS_PER_CU=$b python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | ms)"; done
  line="$line default $(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | ms)"
  echo "$line" | tee -a $OUT
done
echo "rep $rep world 1: default $(python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | ms)  bpc4 $(RVPT_HIP_BLOCKS_PER_CU=4 python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | ms)" | tee -a $OUT
done
