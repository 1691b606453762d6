#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_full
timeout 900 python bench.py > gpurun_out/r06_full/bench.json 2> gpurun_out/r06_full/bench.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r06_full/bench.json; tail -3 gpurun_out/r06_full/bench.err
python scripts/dp_cost.py > gpurun_out/r06_full/dp_cost.txt 2>&1; grep -v "^$" gpurun_out/r06_full/dp_cost.txt | cut -c1-250 | tail -20
