cd $GRAFT_REPO_ROOT
C="--n-background 60000 --n-road 20000 --traversals 3 --objects 4 --width 320 --height 200 --steps 45 --refine-every 20 --reps 1 --only fused --shipped --optimizer fused --visfirst"
python scripts/mtgs_like_train.py $C 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400
python scripts/mtgs_like_train.py $C --row-lazy 2>&1 | grep -v amdgpu | tail -4 | cut -c1-400
for x in "" "--row-lazy"; do
python scripts/mtgs_like_train.py --shipped --visfirst --optimizer fused --graph --reps 20 $x 2>&1 | grep -v amdgpu | tail -2 | cut -c1-400
done
