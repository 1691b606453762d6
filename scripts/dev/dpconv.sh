cd $GRAFT_REPO_ROOT
run() { echo "== $*"; python scripts/mtgs_like_train.py --accumulate 8 --traversals 8 --width 960 --height 540 --reps 1 --only fused --shipped --converge --clear-radius 12 "$@" 2>&1 | grep -E "refine|loss:|converge|Error|error" | cut -c1-200; }
run --steps 120 --refine-every 20 --densify-from 50 --grad-thresh 1e-3
run --steps 120 --refine-every 20 --densify-from 50 --grad-thresh 3e-3
run --steps 120 --refine-every 20 --densify-from 50 --grad-thresh 1e-2
run --steps 140 --refine-every 40 --densify-from 30 --grad-thresh 3e-3
