# Development: a longer training run (400 steps, 7 refinements, object nodes) with and without the row-lazy optimizer
cd $GRAFT_REPO_ROOT
C="--n-background 60000 --n-road 20000 --traversals 3 --objects 4 --width 320 --height 200 --steps 400 --refine-every 50 --reps 1 --only fused --shipped --optimizer fused --visfirst"
python scripts/mtgs_like_train.py $C 2>&1 | grep -E "refine|loss:|timing" | cut -c1-160
echo ---- row-lazy
python scripts/mtgs_like_train.py $C --row-lazy 2>&1 | grep -E "refine|loss:|timing" | cut -c1-160
