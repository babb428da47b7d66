#!/bin/bash
# Run on the GPU box: A/B of library builds under build/exp/ on ONE box, headline configuration (K = 20 and K = 200), two repetitions.  usage: tools/archive/ab_libs.sh name1 name2 ...  -> gpurun_out/ab_libs.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_libs.txt
: > $OUT
val() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null || echo FAILED; }
for rep in 1 2; do for name in "$@"; do
  lib=$REPO/build/exp/librvpt_$name.so; [ "$name" = default ] && lib=$REPO/rvpt_amd/librvpt_hip.so
  a=$(RVPT_HIP_LIB=$lib python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)
  b=$(RVPT_HIP_LIB=$lib python $REPO/bench.py --no-cpu-baseline 2>/dev/null | val)
  echo "rep $rep $name K20 $a K200 $b" | tee -a $OUT
done; done
