#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_dp.py -x -q 2>&1 | tail -12
) > gpurun_out/exp_train.log 2>&1
