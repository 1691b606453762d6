#!/bin/bash
# Development: rocprofv3 kernel-trace summary of scripts/fbench.py (the fused rasterization alone) -> gpurun_out/prof_fbench[_$1]/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_fbench${1:+_$1}
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/fbench.py --reps 20 > $OUT/log.txt 2>&1
echo "rc=$?"
python3 scripts/show_stats.py $(basename $OUT) 26
