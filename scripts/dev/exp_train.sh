#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r04_final.json
cut -c1-400 gpurun_out/bench_r04_final.json
