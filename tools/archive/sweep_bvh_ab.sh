cd $GRAFT_REPO_ROOT
for cfg in "32 0" "32 64" "8 256" "8 512" "6 512" "32 0" "8 512"; do set -- $cfg
  STACKS=$1 TOPS=$2 BPCS=3 bash tools/archive/sweep_bvh_top.sh
done
