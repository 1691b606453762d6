#!/bin/bash
# Development: kernel trace of bench.py with the background fill at MTGS_PREFILL_BLOCKS (exported by the caller)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_prefill
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-also --no-tight > $OUT/bench.log 2>&1
echo "rc=$?"
