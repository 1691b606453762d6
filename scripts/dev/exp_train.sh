#!/bin/bash
mkdir -p gpurun_out
(
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_mtgs_contract.py -x -q -k "graph" 2>&1 | tail -15
) > gpurun_out/exp_train.log 2>&1
