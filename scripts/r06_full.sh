#!/bin/bash
# Round 6: the whole -m gpu suite + the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_full
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r06_full/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_full/pytest_gpu.log
tail -12 gpurun_out/r06_full/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06_full/bench.json 2> gpurun_out/r06_full/bench.err
echo "bench rc=$?"; cut -c1-700 gpurun_out/r06_full/bench.json; tail -3 gpurun_out/r06_full/bench.err
