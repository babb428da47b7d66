cd $GRAFT_REPO_ROOT
for b in 4 5 7 8 10 20; do
  r=$(python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  r2=$(python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "batch $b : K20 $r  again $r2"
done
