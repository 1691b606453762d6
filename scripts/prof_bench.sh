#!/bin/bash
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_bench/ (copied to profiles/ by scripts/collect_profiles.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-also --no-tight > $OUT/bench.log 2>&1
echo "rc=$?"
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-150 | head -40'
