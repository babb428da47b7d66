# 8 frames per launch: frames in flight x work-groups per CU, HBM-resident BVH kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"3 3" "3 2" "4 2" "6 2" "6 1" "4 3"}; do
  set -- $cfg
  export RVPT_HIP_FRAMES_IN_FLIGHT=$1 RVPT_HIP_BLOCKS_PER_CU=$2
  a=$(python bench.py --scene cornell --aa 4 --traversal bvh --steps 48 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(python bench.py --scene heightfield --traversal bvh --steps 160 --warmup 16 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "in flight $1, work-groups per CU $2 : cornell $a   terrain $b"
done
