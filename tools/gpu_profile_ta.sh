#!/bin/bash
# Run on the GPU box: the texture-addresser / vector-L1 side of a BVH configuration (is the walk bound by the path its per-lane 16-byte loads take?).
# Usage: [STEPS=8] tools/gpu_profile_ta.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/ta_*.csv
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --steps ${STEPS:-8} --warmup ${STEPS:-8} --ramp-seconds 0 $*"
run() { local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 300 rocprofv3 "$@" -d /tmp/rp_$name -o $name --output-format csv -- $BENCH > $OUT/$name.bench.log 2>&1
  find /tmp/rp_$name -name '*counter_collection.csv' | while read f; do (head -1 $f; grep -E 'trace_' $f) > $OUT/$(basename $f); done
}
run ta_busy --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run ta_sq --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ta_tcp --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_TAGRAM0_REQ_sum
run ta_tcp2 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
ls $OUT
