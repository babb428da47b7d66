#!/bin/bash
# Run on the GPU box: one rank's share of an 8-way partition of the BVH workloads (C4 geometry, C3) at the driver's --steps 20 --warmup 5, by
# launch shape (one 20-frame launch / 10+10 / 7+7+6 / 5x4 / 20x1) and work-groups per CU.  -> gpurun_out/share_shapes.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/share_shapes.txt
mkdir -p $REPO/gpurun_out; : > $OUT
ms() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('ms_per_frame_wall', d.get('ms_per_step')), d.get('launch'))" 2>/dev/null || echo FAILED; }
for cfg in "c4geo --scene heightfield --traversal bvh" "c3 --scene cornell --aa 4 --traversal bvh"; do
  set -- $cfg; name=$1; shift
  for rank in ${RANKS:-2 5}; do
    for shape in 20 10,10 7,7,6 5,5,5,5 4,4,4,4,4; do
      for bpc in 0 2 3 4 6; do
        envs=(); [ $bpc != 0 ] && envs=(RVPT_HIP_BLOCKS_PER_CU=$bpc)
        echo "$name rank $rank of 8 launches $shape bpc $bpc: $(env "${envs[@]}" X=1 python $REPO/bench.py "$@" --steps 20 --warmup 5 --emulate-world 8 --emulate-rank $rank --batch 64 --launches $shape 2>/dev/null | ms)" | tee -a $OUT
      done
    done
  done
done
