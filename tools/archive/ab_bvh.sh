# A/B of library builds on ONE box (box-to-box spread of the BVH kernel is ~10 %): usage  tools/archive/ab_bvh.sh a.so b.so ...   (paths relative to the repo)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for lib in "$@"; do
  a=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --scene cornell --aa 4 --traversal bvh --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --scene heightfield --traversal bvh --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$lib : cornell $a   heightfield $b"
done; done
