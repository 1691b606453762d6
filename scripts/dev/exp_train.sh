#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_dp.py -x -q -k "rows_into or routes or sparse_exchange" 2>&1 | tail -5
echo "=== dp_cost 1080p"; timeout 900 python scripts/dp_cost.py --no-render-leg 2>&1 | grep -v amdgpu.ids
) > gpurun_out/exp_train.log 2>&1
