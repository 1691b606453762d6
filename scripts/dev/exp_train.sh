#!/bin/bash
# scratch: one-graph (device traversal) tests + timing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_adam.py -x -q -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -m gpu -k "one_graph or through_graphs" 2>&1 | tail -15
C="--shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12"
for T in 3 8; do
for og in "" "--one-graph"; do
echo "== T=$T $og"
timeout 900 python scripts/mtgs_like_train.py --n-background 1600000 --n-road 400000 --traversals $T --steps 1200 --refine-every 100 --densify-from 500 --train-graph $og --steady 1060 1190 $C 2>&1 | grep -E "timing|steady|converge|loss:|Error|error" | tail -6
done; done
