# HBM-resident BVH kernel: leaf batch x refill threshold (work-groups per CU 3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lb in ${LBS:-8 16 24 32}; do for rf in ${RFS:-16 32 48}; do
  export RVPT_HIP_BVH_LEAF_BATCH=$lb RVPT_HIP_BVH_REFILL=$rf
  BPCS=3 bash tools/archive/sweep_bvh.sh | sed "s/^/leaf_batch $lb refill $rf /"
done; done
