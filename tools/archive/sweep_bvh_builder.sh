# builder knob: relative cost of an inner-node visit (0 = the reference's pure area x count rule)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for tc in ${TCS:-0 0.5 1 2 3 4}; do
  export RVPT_BVH_TRAVERSAL_COST=$tc
  BPCS=3 bash tools/archive/sweep_bvh.sh | sed "s/^/traversal_cost $tc /"
done
