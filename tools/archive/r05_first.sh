#!/bin/bash
# round 5, first GPU call: the rectangle cull of the packet kernel — its tests, the headline A/B, then the whole GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "camera_rects or with_and_without_the_rectangles or packet_kernel_equals or full_hd" 2>&1 | tail -5 | tee gpurun_out/r05_rect_tests.txt
for i in 1 2; do
  RVPT_HIP_PACKETS_CULL=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_k20_nocull_$i.json
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05_k20_cull_$i.json
done
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_default_cull.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_k20_*.json"))+["gpurun_out/r05_default_cull.json"]:
    try:
        d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["config"]["lds_bytes_per_block"], d["config"]["grid_blocks"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05_all_tests.txt
