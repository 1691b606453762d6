#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -m gpu -k "dp_rows or configs4 or data_parallel or sparse" 2>&1 | tail -3
(timeout 900 python scripts/dp_cost.py --no-render-leg; timeout 900 python scripts/dp_cost.py --no-render-leg --width 960 --height 540) 2>&1 | grep -v amdgpu.ids > gpurun_out/dp_cost_r04.txt
