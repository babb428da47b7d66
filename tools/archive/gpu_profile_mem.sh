#!/bin/bash
# Memory-pipeline counters of a bench configuration (TA / TCP(vL1D) / L2 request latency), separate --pmc passes.
# Usage: tools/archive/gpu_profile_mem.sh <tag> [bench.py args...]   -> gpurun_out/profmem_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0 --steps ${STEPS:-16} --warmup ${WARMUP:-8} $*"
run() {
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 150 rocprofv3 "$@" -d /tmp/rp_$name -o $name --output-format csv -- $BENCH > $OUT/$name.bench.log 2>&1
  find /tmp/rp_$name -name '*counter_collection.csv' | while read f; do (head -1 $f; grep -E 'trace_' $f) > $OUT/$(basename $f); done
}
# few counters per pass (a TA/TCP block has two to four counter slots; an oversubscribed pass aborts rocprofv3), every pass under a timeout
run ta1 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run ta2 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run sq --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
ls $OUT
