# (historic: the binary-tree camera-packet instances this script measures were retired — apply profiles/r04_exp_campack_binary.patch to reproduce; results: profiles/r04_campack*.txt, r04_ab_wide_resident.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/ab_wide_resident2.txt
: > $OUT
one() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label ${envs[*]} : $(env "${envs[@]}" timeout 300 python $REPO/bench.py --no-cpu-baseline --ramp-seconds 0.5 --traversal bvh "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['grid_blocks'], d['config']['lds_bytes_per_block'])" 2>/dev/null || echo FAILED)" | tee -a $OUT; }
L="--steps 296 --warmup 32"
for st in 2 3 4 5; do for b in 3 4 5; do one wr RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=$st RVPT_HIP_BLOCKS_PER_CU=$b -- $L; done; done
one wr_256 RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=4 -- --width 256 --height 256 --steps 400 --warmup 40
one wr_b1 RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=4 -- $L --batch 1
one wr_b1_bpc3 RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=4 RVPT_HIP_BLOCKS_PER_CU=3 -- $L --batch 1
one campack_stack4 RVPT_HIP_BVH_STACK_LDS=4 -- $L
one wr_k20 RVPT_HIP_BVH_WIDE_RESIDENT=1 RVPT_HIP_BVH_STACK_LDS=4 -- --steps 20 --warmup 5
one campack_k20 X=1 -- --steps 20 --warmup 5
