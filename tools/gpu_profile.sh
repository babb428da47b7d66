#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of the headline bench.
# Usage: [STEPS=20 WARMUP=3] tools/gpu_profile.sh <tag> [bench.py args...]      -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every dispatch of the profiled run carries the same number of frames (warm-up = steps, no clock ramp): the per-dispatch means of the
# counters then belong to that launch shape, which tools/summarize_prof.py records as frames_per_launch
BENCH="python $REPO/bench.py --no-cpu-baseline --no-one-frame-leg --steps ${STEPS:-20} --warmup ${WARMUP:-${STEPS:-20}} --ramp-seconds 0 $*"
# the kernel-trace pass is where DURATIONS come from: its timed region is ${REPS:-12} launches of the same shape (STEPS frames each; --launches pins
# the shape whatever --batch rule applies), each waited for before the next goes out (--serial-launches), so that tools/summarize_prof.py finds >= 10
# launches that overlap no other frame kernel.  Launches queued behind each other on the rotating streams record the queueing as duration, and
# bench.py's three buffer-pre-grow launches run concurrently on three streams: neither is a kernel duration
REPS=${REPS:-12}
SHAPE=$(python3 -c "print(','.join(['${STEPS:-20}'] * $REPS))")
TRACE_BENCH="python $REPO/bench.py --no-cpu-baseline --no-one-frame-leg --steps $((${STEPS:-20} * REPS)) --warmup ${WARMUP:-${STEPS:-20}} --ramp-seconds 0 --batch ${STEPS:-20} --launches $SHAPE --serial-launches $*"
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  local cmd=$BENCH
  [ "$name" = trace ] && cmd=$TRACE_BENCH
  timeout 300 rocprofv3 "$@" -d /tmp/rp_$name -o $name --output-format csv -- $cmd > $OUT/$name.bench.log 2>&1
  find /tmp/rp_$name -name '*.csv' | while read f; do
    b=$(basename $f)
    # keep only the frame kernels' rows (plus header) to stay small
    (head -1 $f; grep -E 'trace_|prepare_triangles|untile|read_rowmajor|blend_accumulate' $f) > $OUT/$b
  done
}
run trace --kernel-trace --stats
run pmc_valu --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
run pmc_wait --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum
ls -la $OUT
