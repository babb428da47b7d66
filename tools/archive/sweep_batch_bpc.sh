cd $GRAFT_REPO_ROOT
for b in 1 2 4 8; do for bpc in 0 2 3 5 6; do
  r=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
  r2=$(RVPT_HIP_BLOCKS_PER_CU=$bpc python bench.py --steps 200 --warmup 20 --batch $b --no-cpu-baseline --ramp-seconds 0.3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "batch $b bpc $bpc : K20 $r  K200 $r2"
done; done
