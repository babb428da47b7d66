# A/B of library builds on ONE box, headline brute-force config: the driver's K = 20 command and the default 200-step run
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for lib in "$@"; do
  a=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  b=$(RVPT_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$lib : K=20 $a   K=200 $b"
done; done
