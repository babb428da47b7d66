# HBM-resident BVH kernel: Cornell (C2 geometry) and the 1M-triangle heightfield (C3) at 1080p over blocks-per-CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
for bpc in ${BPCS:-2 3 4}; do
  a=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --scene cornell --aa 4 --traversal bvh --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  b=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --scene heightfield --traversal bvh --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  echo "bpc $bpc : cornell $a   heightfield $b"
done
