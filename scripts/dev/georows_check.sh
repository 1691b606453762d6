cd $GRAFT_REPO_ROOT
C="--n-background 60000 --n-road 20000 --traversals 3 --width 320 --height 200 --steps 45 --refine-every 20 --reps 1 --only fused --shipped --optimizer fused --visfirst"
python scripts/mtgs_like_train.py $C 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
echo "---- geometry rows"
python scripts/mtgs_like_train.py $C --geometry-rows 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
echo "---- geometry rows + row lazy"
python scripts/mtgs_like_train.py $C --geometry-rows --row-lazy 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300
for x in "--row-lazy" "--row-lazy --geometry-rows"; do
echo "$x: $(python scripts/mtgs_like_train.py --shipped --visfirst --optimizer fused --graph --reps 24 $x 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150)"
done
