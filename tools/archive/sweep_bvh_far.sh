# HBM-resident BVH kernel: far-fetch batch x leaf batch (Cornell 1080p x 4 spp / 1M-triangle terrain)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"1 16" "8 16" "16 16" "24 16" "32 16" "16 24" "16 32" "1 24" "1 32"}; do
  set -- $cfg
  export RVPT_HIP_BVH_FAR_BATCH=$1 RVPT_HIP_BVH_LEAF_BATCH=$2
  a=$(python bench.py --scene cornell --aa 4 --traversal bvh --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  b=$(python bench.py --scene heightfield --traversal bvh --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "far_batch=$1 leaf_batch=$2 : cornell $a   heightfield $b"
done
