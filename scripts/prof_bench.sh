#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench workload -> gpurun_out/bench_prof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/bench_prof && mkdir -p $R/gpurun_out/bench_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bench_prof -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-steps 0 > $R/gpurun_out/bench_prof/log.txt 2>&1
tail -1 $R/gpurun_out/bench_prof/log.txt | cut -c1-200
