#!/bin/bash
# Run on the GPU box: the packet kernel of the headline configuration — work-groups per CU, frames per launch, launches in flight, experiment builds.  -> gpurun_out/packets_sweep.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/packets_sweep.txt
: > $OUT
val() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])" 2>/dev/null || echo FAILED; }
one() {  # label, env..., then -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  a=$(env "${envs[@]}" python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | val)
  b=$(env "${envs[@]}" python $REPO/bench.py --no-cpu-baseline "$@" 2>/dev/null | val)
  echo "$label K20 $a K200 $b" | tee -a $OUT
}
one default X=1 --
one mixed_packets X=1 -- --mixed-packets
for b in 2 3 4 5 6; do one bpc$b RVPT_HIP_BLOCKS_PER_CU=$b --; done
for f in 2 4 6; do one inflight$f RVPT_HIP_FRAMES_IN_FLIGHT=$f --; done
for bt in 1 2 4 16; do one batch$bt X=1 -- --batch $bt; done
one bounce_early RVPT_HIP_LIB=$REPO/build/exp/librvpt_pk_bounce_early.so --
for n in 2 4 8; do
  a=$(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | tail -1)
  b=$(python $REPO/bench.py --steps 20 --warmup 5 --emulate-world $n --mixed-packets 2>/dev/null | tail -1)
  echo "emulate-world $n packets: $a" | tee -a $OUT; echo "emulate-world $n mixed  : $b" | tee -a $OUT
done
one default_again X=1 --
