#!/bin/bash
# Run on the GPU box: the round's evidence.  Part 1 (no argument): the GPU test suite, rocprofv3 passes (kernel trace + PMC) of the headline, the driver's bench
# command three times + the default run.  Part 2 (argument "bvh"): the BVH configurations under rocprofv3.  Part 3 ("table"): every configuration's bench line
# (run AFTER parts 1-2 have been summarised into profiles/pmc_traffic.json: the lines replay it) + every rank's share of an 8-way partition.
# Everything lands under gpurun_out/; tools/summarize_prof.py turns the passes into profiles/.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
if [ "$1" = bvh ]; then
  STEPS=8 REPS=10 timeout 500 tools/gpu_profile.sh c3_wide --scene cornell --aa 4 --traversal bvh > gpurun_out/prof_c3_wide.log 2>&1
  STEPS=8 REPS=10 timeout 500 tools/gpu_profile.sh hf_wide --scene heightfield --traversal bvh > gpurun_out/prof_hf_wide.log 2>&1
  STEPS=8 REPS=10 timeout 400 tools/gpu_profile.sh default_bvh --traversal bvh > gpurun_out/prof_default_bvh.log 2>&1
  ls gpurun_out/prof_c3_wide gpurun_out/prof_hf_wide gpurun_out/prof_default_bvh | head -60
  exit 0
fi
if [ "$1" = table ]; then
  tools/bench_table.sh > gpurun_out/config_table.txt 2>&1
  tail -25 gpurun_out/config_table.txt
  STEPS=20 tools/rank_shares.sh 8 > /dev/null 2>&1
  cat gpurun_out/rank_shares.txt
  exit 0
fi
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/final_tests.log | tail -3 | tee gpurun_out/final_tests.txt
STEPS=20 timeout 500 tools/gpu_profile.sh packets > gpurun_out/prof_packets.log 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_k20_run$i.json; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json
for f in ("bench_k20_run1", "bench_k20_run2", "bench_k20_run3", "bench_default"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_issue"), d["cpu_baseline"]["value"], d.get("cpu_baseline_bvh", {}).get("value"))
PY
