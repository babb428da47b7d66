#!/bin/bash
# LDS / vector-memory / TA pipe counters of one bench configuration, three short --pmc passes; prints per-dispatch means of
# the frame kernel.  RVPT_HIP_LIB selects an experimental library build.  Usage: tools/archive/gpu_profile_pipes.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pipes_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0 --steps ${STEPS:-16} --warmup ${WARMUP:-8} $*"
run() {
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 150 rocprofv3 "$@" -d /tmp/rp_$name -o $name --output-format csv -- $BENCH > $OUT/$name.bench.log 2>&1
  find /tmp/rp_$name -name '*counter_collection.csv' | while read f; do (head -1 $f; grep -E 'trace_' $f) > $OUT/$(basename $f); done
}
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run ta1 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k in sorted(acc):
    per = collections.defaultdict(float)
    for d, v in acc[k]:
        per[d] += v
    vals = sorted(per.values())
    print(f"{k:32s} mean/dispatch {sum(vals)/len(vals):16.0f}   dispatches {len(vals)}")
PY
