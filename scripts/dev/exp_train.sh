#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout 1500 python -m pytest tests/test_gpu_mtgs_contract.py -q -m gpu -k "configs4_eight" > gpurun_out/c4_$i.log 2>&1
echo rc=$?; tail -2 gpurun_out/c4_$i.log
done
