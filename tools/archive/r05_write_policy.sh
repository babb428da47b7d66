#!/bin/bash
# round 5: a nontemporal hint on the 16-byte sample store (build variant -DRV_SAMPLE_STORE_NT=1) — WRITE_SIZE per 8-frame launch and throughput, C3 and C4 geometry
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for scene in "cornell --aa 4" "heightfield"; do
  for lib in default store_nt; do
    [ $lib = default ] && unset RVPT_HIP_LIB || export RVPT_HIP_LIB=$REPO/build/exp/$lib.so
    rm -rf /tmp/rp_w
    timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/rp_w -o w --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 8 --warmup 8 --ramp-seconds 0 --traversal bvh --scene $scene > /dev/null 2>&1
    w=$(find /tmp/rp_w -name '*counter_collection.csv' | head -1 | xargs python3 -c "
import csv,sys
v=[float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'trace_bvh4' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE']
print(f'{sum(v)/len(v)*1024/1e6:.1f} MB per launch over {len(v)} launches')")
    t=$(python $REPO/bench.py --no-cpu-baseline --steps 96 --warmup 16 --traversal bvh --scene $scene 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
    echo "$scene $lib: WRITE_SIZE $w; $t Msamples/s"
  done
done
