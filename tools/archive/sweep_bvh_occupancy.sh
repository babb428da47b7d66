# HBM-resident BVH kernel: resident work-groups per CU with ONE launch in flight (8 frames per launch, so the tail is small):
# (two launches in flight x N work-groups per CU each = 2N resident) does throughput follow the number of waves (latency-bound) or saturate (a unit is full)?
cd ${GRAFT_REPO_ROOT:-/root/repo}
export RVPT_HIP_FRAMES_IN_FLIGHT=${INFLIGHT:-2}
for bpc in ${BPCS:-1 2 3}; do
  a=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --scene cornell --aa 4 --traversal bvh --batch 8 --steps 40 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frames_per_dispatch'])")
  b=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --scene heightfield --traversal bvh --batch 8 --steps 80 --warmup 8 --no-cpu-baseline --ramp-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "work-groups per CU $bpc : cornell $a   heightfield $b"
done
