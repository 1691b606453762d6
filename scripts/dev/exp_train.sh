#!/bin/bash
cd $GRAFT_REPO_ROOT
C="--shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12"
MTGS_REFINE_DEBUG=1 MTGS_TRAIN_DEBUG=1 timeout 900 python scripts/mtgs_like_train.py $C --steps 520 --refine-every 100 --densify-from 250 --train-graph --one-graph 2>&1 | grep -E "refine |step |debug phases|timing" | tail -40
