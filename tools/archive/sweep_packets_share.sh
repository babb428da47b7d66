#!/bin/bash
# Run on the GPU box: rank 0's share of an 8-way (and 4-way) tile partition at the driver's command, packet kernel: work-groups per CU, launches in flight, launch split.  -> gpurun_out/packets_share.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/packets_share.txt
: > $OUT
ms() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_frame_wall', d.get('ms_per_step')))" 2>/dev/null || echo FAILED; }
for n in 8 4; do
  echo "world $n default: $(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | ms)  mixed: $(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n --mixed-packets 2>/dev/null | ms)" | tee -a $OUT
  for b in 1 2 3 4; do echo "world $n bpc$b: $(RVPT_HIP_BLOCKS_PER_CU=$b python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | ms)" | tee -a $OUT; done
  for f in 2 4 6; do echo "world $n inflight$f: $(RVPT_HIP_FRAMES_IN_FLIGHT=$f python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | ms)" | tee -a $OUT; done
  echo "world $n one-launch: $(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n --no-launch-split 2>/dev/null | ms)" | tee -a $OUT
  for b in 2 3; do echo "world $n one-launch bpc$b: $(RVPT_HIP_BLOCKS_PER_CU=$b python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n --no-launch-split 2>/dev/null | ms)" | tee -a $OUT; done
done
echo "world 1: $(python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | ms)" | tee -a $OUT
