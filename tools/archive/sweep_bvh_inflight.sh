cd $GRAFT_REPO_ROOT
for f in 2 3 4; do for bpc in 2 3 6; do
  export RVPT_HIP_FRAMES_IN_FLIGHT=$f
  BPCS=$bpc bash tools/archive/sweep_bvh.sh | sed "s/^/in_flight $f /"
done; done
