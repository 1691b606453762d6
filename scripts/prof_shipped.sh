#!/bin/bash
# rocprofv3 kernel-trace summary of the fused MTGS-like iteration with the shipped option set -> gpurun_out/prof_shipped/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_shipped && mkdir -p $R/gpurun_out/prof_shipped
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_shipped -o shipped -- python3 $R/scripts/mtgs_like_train.py --shipped --only fused --reps 10 > $R/gpurun_out/prof_shipped/log.txt 2>&1
tail -1 $R/gpurun_out/prof_shipped/log.txt | cut -c1-200
