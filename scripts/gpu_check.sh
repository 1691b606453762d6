#!/bin/bash
# Development: the GPU test suite + a short bench, logs under gpurun_out/ (run through gpurun).
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-steps 0 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "bench rc=$?"
cat gpurun_out/bench_quick.json | cut -c1-600
tail -5 gpurun_out/bench_quick.err
