#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kb && mkdir -p $R/gpurun_out/kb
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kb -o kb -- python3 $R/scripts/kbench.py --reps 5 > $R/gpurun_out/kb/log.txt 2>&1
find $R/gpurun_out/kb -name "*stats*" | head
