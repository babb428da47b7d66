# HBM-resident BVH kernel: LDS stack levels x tree-top nodes (Cornell 1080p x 4 spp / 1M-triangle terrain); LIB selects a library build
cd ${GRAFT_REPO_ROOT:-/root/repo}
[ -n "${LIB:-}" ] && export RVPT_HIP_LIB=$PWD/$LIB
IFS=';'; for cfg in ${CFGS:-8 256;7 256;6 320;8 192;6 256}; do
  IFS=' '; set -- $cfg
  a=$(RVPT_HIP_BVH_STACK_LDS=$1 RVPT_HIP_BVH_TOP_NODES=$2 python bench.py --scene cornell --aa 4 --traversal bvh --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['lds_bytes_per_block'])")
  b=$(RVPT_HIP_BVH_STACK_LDS=$1 RVPT_HIP_BVH_TOP_NODES=$2 python bench.py --scene heightfield --traversal bvh --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "${LIB:-in-tree} S=$1 top=$2 : cornell $a   heightfield $b"
done
