#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace + stats only (no counters) of one bench configuration.  -> gpurun_out/trace_<tag>/
# usage: [STEPS=16 WARMUP=8] tools/archive/prof_trace_only.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$TAG -o t --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps ${STEPS:-16} --warmup ${WARMUP:-8} --ramp-seconds 0 "$@" > $OUT/bench.log 2>&1
find /tmp/rp_$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
# per-launch durations of the pipeline's kernels, in launch order (the tail of the trace: the timed steps)
f=$(find /tmp/rp_$TAG -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/sequence.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("wf_", "trace_", "blend_"))]
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows[-400:]:
    name = r["Kernel_Name"].split("(")[0].replace("rv::", "").replace("void ", "")
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us  q{r.get("Queue_Id", "?")}  {name}')
PY
tail -3 $OUT/bench.log; cat $OUT/kernel_stats.csv | cut -c1-160
