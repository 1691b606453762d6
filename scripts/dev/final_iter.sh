# Development: the headline numbers of the MTGS-like iteration (960x540, shipped options, 2M Gaussians) as one HIP graph
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(python scripts/mtgs_like_train.py --shipped --graph --reps 24 $2 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140)"; }
run "dense node path, no optimizer" ""
run "visibility first, no optimizer" "--visfirst"
run "visibility first + FusedAdam (every row) T=3" "--visfirst --optimizer fused"
run "visibility first + FusedAdam --row-lazy T=3" "--visfirst --optimizer fused --row-lazy"
run "visibility first + FusedAdam (every row) T=8" "--visfirst --optimizer fused --traversals 8"
run "visibility first + FusedAdam --row-lazy T=8" "--visfirst --optimizer fused --row-lazy --traversals 8"
