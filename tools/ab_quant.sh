#!/bin/bash
# A/B of the quantised-node walk (trace_bvh4q) against the exact 128-byte nodes: C3, C4 geometry (and with ALL=1 C5 and the driver's shapes) at bench shapes.
# Libraries built by hand beside the default one: librvpt_hip_q6.so (-DRV_BVH4Q_MIN_WAVES=6: 80 VGPRs, six spilled)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/ab_quant.txt; : > $OUT
run() { label=$1; shift
  v=$(env "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$label $v" | tee -a $OUT; }
C3="--scene cornell --aa 4 --traversal bvh --steps 96 --warmup 16 --no-cpu-baseline"
C4="--scene heightfield --traversal bvh --steps 96 --warmup 16 --no-cpu-baseline"
for rep in 1 2; do for cfg in C3 C4; do args=${!cfg}
  run "$cfg exact" RVPT_HIP_BVH_QUANT=0 python bench.py $args
  run "$cfg quant5" RVPT_HIP_BVH_QUANT=1 python bench.py $args
  for v in ${LIBS:-q6}; do run "$cfg quant_$v" RVPT_HIP_BVH_QUANT=1 RVPT_HIP_LIB=$PWD/rvpt_amd/librvpt_hip_$v.so python bench.py $args; done
done; done
