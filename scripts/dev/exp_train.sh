#!/bin/bash
mkdir -p gpurun_out
C="--shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12"
(
echo "=== tests"; timeout 2400 python -m pytest tests/test_gpu_adam.py -x -q 2>&1 | tail -3
timeout 900 python tests/fuzz_rowlazy.py --cases 150 2>&1 | tail -2
echo "=== graph bench"; timeout 600 python scripts/mtgs_like_train.py --shipped --graph --reps 96 --visfirst --optimizer fused --row-lazy --geometry-rows 2>&1 | tail -1 | cut -c1-120
timeout 600 python scripts/mtgs_like_train.py --shipped --graph --reps 96 --visfirst --optimizer fused --row-lazy --geometry-rows --traversals 8 2>&1 | tail -1 | cut -c1-120
echo "=== 2M T=3 graph training"; timeout 900 python scripts/mtgs_like_train.py $C --steps 600 --refine-every 100 --densify-from 250 --steady 60 260 --train-graph 2>&1 | grep -E "timing|steady|converge"
echo "=== 2M T=8 graph training"; timeout 900 python scripts/mtgs_like_train.py $C --traversals 8 --steps 1200 --refine-every 100 --densify-from 500 --steady 100 500 --train-graph 2>&1 | grep -E "timing|steady|converge"
) > gpurun_out/exp_train.log 2>&1
