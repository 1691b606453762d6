#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
C="--shipped --visfirst --optimizer fused --row-lazy --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12 --objects 100 --traversals 8"
( echo "--- 2M Gaussians + 100 rigid objects, eight traversals, one graph per stretch (after the LIST scheduling fix)"
timeout 1200 python scripts/mtgs_like_train.py $C --steps 700 --refine-every 100 --densify-from 300 --steady 100 280 --train-graph --one-graph 2>&1 | grep -E "timing|steady|converge|Error|error|Traceback" | tail -4 ) | tee gpurun_out/training_objects2.txt
