#!/bin/bash
# round 5: the bounce cull of the packet kernel — its tests, the headline A/B, the whole GPU suite with its summary kept
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "camera_rects or with_and_without_the_rectangles or packet_kernel_equals or full_hd or bounce_cull" > gpurun_out/r05b_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r05b_tests.log | tail -5
for i in 1 2; do
  RVPT_HIP_PACKETS_BOUNCE_CULL=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05b_k20_nobounce_$i.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05b_k20_both_$i.json
done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05b_default_both.json
RVPT_HIP_BLOCKS_PER_CU=4 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05b_k20_both_bpc4.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05b_*.json")):
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["config"]["lds_bytes_per_block"], d["config"]["grid_blocks"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05b_all_tests.log 2>&1
grep -E "passed|failed|rror" gpurun_out/r05b_all_tests.log | tail -5
