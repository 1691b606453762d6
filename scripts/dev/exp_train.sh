#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 3000 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -k "dp_rows or eight_traversals" 2>&1 | tail -30
echo "=== dp_cost 1080p"; timeout 900 python scripts/dp_cost.py 2>&1 | grep -v amdgpu.ids
echo "=== dp_cost 540p"; timeout 900 python scripts/dp_cost.py --width 960 --height 540 2>&1 | grep -v amdgpu.ids
) > gpurun_out/exp_train.log 2>&1
