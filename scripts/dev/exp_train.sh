#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
/usr/bin/time -v timeout 900 python bench.py 2> gpurun_out/bench_time.txt | tail -1 > gpurun_out/bench_r04_final.json
grep -E "Elapsed|Maximum resident" gpurun_out/bench_time.txt
cut -c1-300 gpurun_out/bench_r04_final.json
