# one frame per launch (the interactive case: the camera moves, nothing to batch): frames in flight x work-groups per CU, HBM-resident BVH kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in ${FS:-3 4 6}; do for bpc in ${BPCS:-2 3}; do
  export RVPT_HIP_FRAMES_IN_FLIGHT=$f RVPT_HIP_BLOCKS_PER_CU=$bpc
  a=$(python bench.py --scene cornell --aa 4 --traversal bvh --batch 1 --steps 48 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(python bench.py --scene heightfield --traversal bvh --batch 1 --steps 160 --warmup 16 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  c=$(python bench.py --traversal bvh --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "in flight $f, work-groups per CU $bpc : cornell $a   terrain $b   default scene (LDS-resident) $c"
done; done
