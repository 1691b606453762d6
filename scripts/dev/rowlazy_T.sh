# Development: the MTGS-like iteration as one HIP graph (960x540, shipped options, visibility first, fused optimizer) per optimizer mode
cd $GRAFT_REPO_ROOT
for T in 3 8; do for x in "" "--lazy-adam" "--row-lazy" "--row-lazy --dense-normals"; do
echo "T=$T $x: $(python scripts/mtgs_like_train.py --shipped --visfirst --optimizer fused --graph --reps 24 --traversals $T $x 2>&1 | grep -v amdgpu | tail -1 | cut -c1-130)"
done; done
