#!/bin/bash
# Run on the GPU box: EVERY rank's share of an N-way tile partition, one after the other on one GPU (bench.py --emulate-world N --emulate-rank r,
# the driver's exact command otherwise), for the headline configuration, C3 and C4 geometry.  The N-GPU time of a step is the MAX over ranks:
# max / mean is the load imbalance of the ownership rule, N=1 time / max-rank time the strong-scaling forecast (no gather, no per-rank host
# overhead: those are added in DESIGN.md 7).  -> gpurun_out/rank_shares.txt
# usage: [STEPS=20] [CONFIGS="headline c3 c4geo"] tools/rank_shares.sh [worlds, default "2 4 8"]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/rank_shares.txt
mkdir -p $REPO/gpurun_out; : > $OUT
WORLDS=${1:-"2 4 8"}
share() {  # label, bench args...
  label=$1; shift
  base=$(python $REPO/bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "$label N=1 ms_per_frame $base" | tee -a $OUT
  for n in $WORLDS; do
    vals=""
    for ((r=0; r<n; r++)); do
      v=$(python $REPO/bench.py "$@" --emulate-world $n --emulate-rank $r 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_frame_wall'], d['segments_per_sample'], d['kernel_ms'])")
      vals="$vals $v;"
    done
    python - "$label" "$n" "$base" "$vals" <<'PY' | tee -a $OUT
import sys
label, n, base, vals = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
rows = [v.split() for v in vals.split(";") if v.strip()]
ms = [float(r[0]) for r in rows]; seg = [float(r[1]) for r in rows]; km = [float(r[2]) for r in rows]
print(f"{label} N={n} per-rank ms_per_frame {' '.join(f'{m:.5f}' for m in ms)}  segments/sample {' '.join(f'{s:.3f}' for s in seg)}  kernel ms of the launch {' '.join(f'{k:.3f}' for k in km)}")
print(f"{label} N={n} max {max(ms):.5f} mean {sum(ms)/len(ms):.5f} max/mean {max(ms)/(sum(ms)/len(ms)):.3f}  forecast N=1/max {base/max(ms):.2f}x of {n}")
PY
  done
}
CONFIGS=${CONFIGS:-"headline c3 c4geo"}
for cfg in $CONFIGS; do case $cfg in
  headline) share headline --steps ${STEPS:-20} --warmup 5 ;;
  c3) share c3 --scene cornell --aa 4 --traversal bvh --steps ${STEPS:-20} --warmup 5 ;;
  c4geo) share c4geo --scene heightfield --traversal bvh --steps ${STEPS:-20} --warmup 5 ;;
esac; done
