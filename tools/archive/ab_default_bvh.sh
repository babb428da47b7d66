# A/B of library builds on ONE box, LDS-resident BVH kernel (default scene): usage tools/archive/ab_default_bvh.sh a.so b.so ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for lib in "$@"; do
  a=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --traversal bvh --steps 296 --warmup 32 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --width 256 --height 256 --traversal bvh --steps 400 --warmup 40 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  c=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --steps 100 --warmup 20 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$lib : default-scene BVH 1080p $a   256x256 $b   brute 1080p $c"
done; done
