# HBM-resident BVH kernel with lane-cooperative record fetches: LDS stack levels x tree-top nodes (Cornell 1080p x 4 spp / 1M-triangle terrain)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"8 256" "6 256" "8 128" "6 128" "6 192" "5 256"}; do
  set -- $cfg
  a=$(RVPT_HIP_BVH_STACK_LDS=$1 RVPT_HIP_BVH_TOP_NODES=$2 python bench.py --scene cornell --aa 4 --traversal bvh --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  b=$(RVPT_HIP_BVH_STACK_LDS=$1 RVPT_HIP_BVH_TOP_NODES=$2 python bench.py --scene heightfield --traversal bvh --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "S=$1 top=$2 : cornell $a   heightfield $b"
done
