#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 3000 python -m pytest tests/test_gpu_adam.py -x -q 2>&1 | tail -3
echo "=== graph bench"; timeout 600 python scripts/mtgs_like_train.py --shipped --graph --reps 96 --visfirst --optimizer fused --row-lazy --geometry-rows 2>&1 | tail -1 | cut -c1-120
timeout 600 python scripts/adam_bench.py 2>&1 | tail -12
) > gpurun_out/exp_train.log 2>&1
