# HBM-resident BVH kernel: stack levels kept in LDS x nodes of the tree top copied into LDS x work-groups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in ${STACKS:-4 6 8 12}; do for top in ${TOPS:-64 256 512}; do for bpc in ${BPCS:-3}; do
  export RVPT_HIP_BVH_TOP_NODES=$top RVPT_HIP_BVH_STACK_LDS=$s
  BPCS=$bpc bash tools/archive/sweep_bvh.sh | sed "s/^/stack $s top $top /"
done; done; done
