#!/bin/bash
# Development: correctness of the bin3 path on the fused / parity tests, then a kernel trace of the headline bench.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/bin3_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/bin3_pytest.log
tail -25 gpurun_out/bin3_pytest.log
bash scripts/prof_bench.sh 2>&1 | tail -45
tail -3 gpurun_out/prof_bench/bench.log | cut -c1-400
