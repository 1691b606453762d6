#!/bin/bash
mkdir -p gpurun_out
C="--shipped --visfirst --optimizer fused --row-lazy --geometry-rows --only fused --reps 1 --converge --grad-thresh 1e-3 --clear-radius 12"
(
echo "=== tests"; timeout 2400 python -m pytest tests/test_gpu_mtgs_contract.py tests/test_gpu_densify.py -x -q 2>&1 | tail -5
echo "=== 2M T=3 graph training"; timeout 900 python scripts/mtgs_like_train.py $C --steps 1000 --refine-every 100 --densify-from 250 --steady 60 260 --train-graph 2>&1 | grep -E "timing|steady|converge|Error|error"
MTGS_TRAIN_DEBUG=1 timeout 900 python scripts/mtgs_like_train.py $C --steps 1000 --refine-every 100 --densify-from 250 --train-graph 2>&1 | grep -E "debug|timing"
echo "=== 2M T=3 eager"; timeout 900 python scripts/mtgs_like_train.py $C --steps 1000 --refine-every 100 --densify-from 250 2>&1 | grep -E "timing|converge"
) > gpurun_out/exp_train.log 2>&1
